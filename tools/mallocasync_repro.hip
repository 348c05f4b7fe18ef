// Does stream-ordered pool memory (hipMallocAsync / hipFreeAsync, ROCm 7.2) ever reach its next owner while the previous owner's work is
// still writing it?  Round 3 saw "a previous owner's writes landing after ours" late in a long process and bisected it to the allocation
// call of include/giddy/stream_scratch.hpp; this is the same traffic WITHOUT the library: several streams, each repeatedly
//   allocate (sizes vary) -> copy a per-owner pattern in from pinned memory -> verify on the device -> scribble slowly -> free,
// so that a block freed on one stream (its scribble still running) is what the pool has to offer the next allocation on another.
//   hipcc --offload-arch=gfx950 -O2 tools/mallocasync_repro.hip -o tools/mallocasync_repro.bin && tools/mallocasync_repro.bin [iterations [streams [malloc]]]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::printf("%s: %s\n", #call, hipGetErrorString(e_)); return 2; } } while (0)

__global__ void verify(const unsigned* block, size_t words, unsigned pattern, unsigned long long* mismatches)
{
	unsigned long long bad = 0;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t) gridDim.x * blockDim.x) { bad += block[i] != pattern; }
	if (bad) { atomicAdd(mismatches, bad); }
}
__global__ void scribble_slowly(unsigned* block, size_t words, unsigned garbage, int rounds)
{
	for (int r = 0; r < rounds; r++) {
		for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t) gridDim.x * blockDim.x) { block[i] = garbage + r; }
	}
}

int main(int argc, char** argv)
{
	const int iterations = argc > 1 ? std::atoi(argv[1]) : 2000, num_streams = argc > 2 ? std::atoi(argv[2]) : 4;
	const bool pool = argc <= 3 || argv[3][0] != 'm';   // third argument "malloc": the CONTROL — the same traffic on hipMalloc / hipFree blocks
	const size_t most_words = (size_t) 16 << 20;   // 64 MiB
	std::vector<hipStream_t> streams(num_streams);
	std::vector<unsigned*> patterns(num_streams);
	unsigned long long* mismatches;
	OK(hipMalloc(&mismatches, sizeof *mismatches)); OK(hipMemset(mismatches, 0, sizeof *mismatches));
	for (int s = 0; s < num_streams; s++) { OK(hipStreamCreateWithFlags(&streams[s], hipStreamNonBlocking)); OK(hipHostMalloc(&patterns[s], (most_words + 4096) * 4, hipHostMallocDefault)); }
	unsigned long long seed = 12345;
	for (int it = 0; it < iterations; it++) {
		const int s = it % num_streams;
		seed = seed * 6364136223846793005ull + 1442695040888963407ull;
		const size_t words = ((size_t) 1 << (12 + (seed >> 33) % 13)) + ((seed >> 20) & 0xFF0);   // 16 KiB ... 64 MiB, ragged
		const unsigned pattern = 0x10000u * (unsigned) (it + 1) + (unsigned) s;
		OK(hipStreamSynchronize(streams[s]));   // (the pinned pattern buffer of this stream is reused)
		for (size_t i = 0; i < words; i++) { patterns[s][i] = pattern; }
		unsigned* block = nullptr;
		if (pool) { OK(hipMallocAsync(reinterpret_cast<void**>(&block), words * 4, streams[s])); } else { OK(hipMalloc(reinterpret_cast<void**>(&block), words * 4)); }
		OK(hipMemcpyAsync(block, patterns[s], words * 4, hipMemcpyHostToDevice, streams[s]));
		verify<<<256, 256, 0, streams[s]>>>(block, words, pattern, mismatches);
		scribble_slowly<<<64, 256, 0, streams[s]>>>(block, words, 0xDEAD0000u + it, 8);
		if (pool) { OK(hipFreeAsync(block, streams[s])); } else { OK(hipStreamSynchronize(streams[s])); OK(hipFree(block)); }
	}
	OK(hipDeviceSynchronize());
	unsigned long long bad = 0;
	OK(hipMemcpy(&bad, mismatches, sizeof bad, hipMemcpyDeviceToHost));
	std::printf("%s: %d allocations over %d streams, %llu words read back different from what their owner had just copied in%s\n",
		pool ? "hipMallocAsync / hipFreeAsync" : "control (hipMalloc / hipFree)", iterations, num_streams, bad, bad ? " — STALE CONTENTS REPRODUCED" : " — not reproduced");
	return bad != 0;
}
