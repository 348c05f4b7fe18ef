// Does stream-ordered pool memory (hipMallocAsync / hipFreeAsync, ROCm 7.2) ever reach its next owner while the previous owner's work is
// still writing it?  Round 3 saw "a previous owner's writes landing after ours" late in a long process and bisected it to the allocation
// call of include/giddy/stream_scratch.hpp; this is the same traffic WITHOUT the library: several streams, each repeatedly
//   allocate (sizes vary) -> copy a per-owner pattern in from pinned memory -> verify on the device -> scribble slowly -> free,
// so that a block freed on one stream (its scribble still running) is what the pool has to offer the next allocation on another.
//   hipcc --offload-arch=gfx950 -O2 tools/mallocasync_repro.hip -o tools/mallocasync_repro.bin && tools/mallocasync_repro.bin [iterations [streams [malloc|reuse|wait|kernel|threshold|coherent]]]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define OK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { std::printf("%s: %s\n", #call, hipGetErrorString(e_)); return 2; } } while (0)

struct seen_wrong { unsigned long long lowest, highest, count; unsigned value, pad; };
__global__ void describe(const unsigned* block, size_t words, unsigned pattern, seen_wrong* record)
{
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t) gridDim.x * blockDim.x) {
		const unsigned seen = block[i];
		if (seen != pattern) { atomicMin(&record->lowest, (unsigned long long) i); atomicMax(&record->highest, (unsigned long long) i); atomicAdd(&record->count, 1ull); record->value = seen; }
	}
}
template <bool AroundTheCaches = false>
__global__ void verify(const unsigned* block, size_t words, unsigned pattern, unsigned long long* mismatches)
{
	unsigned long long bad = 0;
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t) gridDim.x * blockDim.x) {
		// (AroundTheCaches: a system-scope load — `global_load_dword ... sc0 sc1` — which no L2 of any XCD may answer from a line it holds)
		const unsigned seen = AroundTheCaches ? __hip_atomic_load(block + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : block[i];
		bad += seen != pattern;
	}
	if (bad) { atomicAdd(mismatches, bad); }
}
__global__ void fill(unsigned* block, size_t words, unsigned pattern)
{
	for (size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t) gridDim.x * blockDim.x) { block[i] = pattern; }
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
	const char mode = argc > 3 ? argv[3][0] : 'p';
	const bool pool = mode != 'm' && mode != 'r';   // ('d'escribe, 'c'oherent, 'w'ait, 'k'ernel, 't'hreshold: all on the pool)   // third argument "malloc": the CONTROL — the same traffic on hipMalloc / hipFree blocks
	// round 5, a second reading — what separates "the pool hands out memory too early" from "a copy into memory that a kernel has just written is not seen by the next kernel":
	//   "reuse": ONE hipMalloc block per stream, used by every iteration (the same addresses again and again, as from the pool, but no allocator in the picture);
	//   "wait":  the pool, with a hipStreamSynchronize between the copy and the verifying kernel (were it an ordering problem of copy and kernel, this hides it);
	//   "kernel": the pool, the pattern written by a KERNEL instead of a copy engine (were it the copy path, this is clean)
	//   "threshold": the pool with its release threshold at the maximum, i.e. it never hands memory back to the system at a synchronisation point (the default threshold is 0:
	//            every hipStreamSynchronize with nothing in use trims the pool, and the next allocation maps memory anew — very likely at the same virtual addresses)
	//   "coherent": the pool, the verifying kernel reads with system-scope loads (were it lines of the previous owner's writes still held by an XCD's L2, this is clean)
	const bool coherent_verify = mode == 'c';
	//   "describe": the pool; of the first iterations that read something wrong, WHICH words and WHAT they hold
	const bool describing = mode == 'd' || argc > 4;   // (or a fourth argument after any other mode)
	seen_wrong* records = nullptr; int described = 0;
	if (describing) { OK(hipHostMalloc(&records, 64 * sizeof(seen_wrong), hipHostMallocDefault)); }
	const bool reuse = mode == 'r', wait_after_copy = mode == 'w', fill_by_kernel = mode == 'k', never_trim = mode == 't';
	if (never_trim) {
		hipMemPool_t default_pool; OK(hipDeviceGetDefaultMemPool(&default_pool, 0));
		uint64_t threshold = ~0ull; OK(hipMemPoolSetAttribute(default_pool, hipMemPoolAttrReleaseThreshold, &threshold));
	}
	const size_t most_words = (size_t) 16 << 20;   // 64 MiB
	std::vector<hipStream_t> streams(num_streams);
	std::vector<unsigned*> patterns(num_streams), reused(num_streams, nullptr);
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
		if (pool) { OK(hipMallocAsync(reinterpret_cast<void**>(&block), words * 4, streams[s])); }
		else if (reuse) { if (reused[s] == nullptr) { OK(hipMalloc(reinterpret_cast<void**>(&reused[s]), (most_words + 4096) * 4)); } block = reused[s]; }
		else { OK(hipMalloc(reinterpret_cast<void**>(&block), words * 4)); }
		if (fill_by_kernel) { fill<<<256, 256, 0, streams[s]>>>(block, words, pattern); }
		else { OK(hipMemcpyAsync(block, patterns[s], words * 4, hipMemcpyHostToDevice, streams[s])); }
		if (wait_after_copy) { OK(hipStreamSynchronize(streams[s])); }
		if (describing && described < 12) {
			records[described] = seen_wrong { ~0ull, 0, 0, 0, 0 };
			describe<<<256, 256, 0, streams[s]>>>(block, words, pattern, records + described);
			OK(hipStreamSynchronize(streams[s]));
			if (records[described].count) {
				std::printf("  allocation %d: %zu words at %p, pattern %08x: %llu wrong words, indices %llu ... %llu, one of them holds %08x\n", it, words, (void*) block, pattern,
					records[described].count, records[described].lowest, records[described].highest, records[described].value);
				described++;
			}
		}
		if (coherent_verify) { verify<true><<<256, 256, 0, streams[s]>>>(block, words, pattern, mismatches); }
		else { verify<<<256, 256, 0, streams[s]>>>(block, words, pattern, mismatches); }
		scribble_slowly<<<64, 256, 0, streams[s]>>>(block, words, 0xDEAD0000u + it, 8);
		if (pool) { OK(hipFreeAsync(block, streams[s])); } else if (reuse) { } else { OK(hipStreamSynchronize(streams[s])); OK(hipFree(block)); }
	}
	OK(hipDeviceSynchronize());
	unsigned long long bad = 0;
	OK(hipMemcpy(&bad, mismatches, sizeof bad, hipMemcpyDeviceToHost));
	std::printf("%s: %d allocations over %d streams, %llu words read back different from what their owner had just copied in%s\n",
		reuse ? "one hipMalloc block per stream, reused" : wait_after_copy ? "hipMallocAsync / hipFreeAsync, the host waits between copy and kernel" : fill_by_kernel ? "hipMallocAsync / hipFreeAsync, pattern written by a kernel"
			: coherent_verify ? "hipMallocAsync / hipFreeAsync, verified with system-scope loads"
			: never_trim ? "hipMallocAsync / hipFreeAsync, release threshold at the maximum (the pool never trims)"
			: pool ? "hipMallocAsync / hipFreeAsync" : "control (hipMalloc / hipFree)", iterations, num_streams, bad, bad ? " — STALE CONTENTS REPRODUCED" : " — not reproduced");
	return bad != 0;
}
