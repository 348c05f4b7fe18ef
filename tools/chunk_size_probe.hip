// development aid (GPU box): a bare streaming copy in which every WAVEFRONT takes a contiguous chunk of S bytes (rows of 1 KiB: a 16-byte load and a 16-byte store
// per lane, two rows in flight; 256-thread workgroups, one chunk per wavefront — the shape of the segmented decoders' wave paths).  Is S = 32 KiB special for the
// memory system itself, or only for the decoders?     hipcc --offload-arch=gfx950 -O3 tools/chunk_size_probe.hip -o /tmp/chunk_size_probe && /tmp/chunk_size_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef uint32_t v4 __attribute__((ext_vector_type(4)));

template <unsigned InputDivisor>   // the input is 1 / InputDivisor of the output (a decoder reads less than it writes): the lane's 16 input bytes are re-used
__global__ void __launch_bounds__(256) copy_chunks(v4* __restrict__ out, const v4* __restrict__ in, uint64_t rows, uint64_t rows_per_chunk)
{
	const unsigned lane = threadIdx.x % 64;
	const uint64_t wave = (uint64_t) blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
	const uint64_t num_waves = (uint64_t) gridDim.x * (blockDim.x / 64);
	const uint64_t num_chunks = (rows + rows_per_chunk - 1) / rows_per_chunk;
	for (uint64_t c = wave; c < num_chunks; c += num_waves) {
		const uint64_t first = c * rows_per_chunk, last = first + rows_per_chunk < rows ? first + rows_per_chunk : rows;
		v4 a = in[(first * 64 + lane) / InputDivisor], b = first + 1 < last ? in[((first + 1) * 64 + lane) / InputDivisor] : a;
		for (uint64_t r = first; r < last; r += 2) {
			v4 na = a, nb = b;
			if (r + 2 < last) { na = in[((r + 2) * 64 + lane) / InputDivisor]; }
			if (r + 3 < last) { nb = in[((r + 3) * 64 + lane) / InputDivisor]; }
			a.x += (uint32_t) r; b.y ^= (uint32_t) r;
			__builtin_nontemporal_store(a, out + r * 64 + lane);
			if (r + 1 < last) { __builtin_nontemporal_store(b, out + (r + 1) * 64 + lane); }
			a = na; b = nb;
		}
	}
}

template <unsigned D> void run(v4* out, v4* in, uint64_t rows, const char* what)
{
	for (unsigned rpc : { 8u, 16u, 21u, 24u, 30u, 31u, 32u, 33u, 48u, 64u, 128u }) {
		const uint64_t chunks = (rows + rpc - 1) / rpc;
		const unsigned grid = (unsigned) ((chunks + 3) / 4);
		hipEvent_t e0, e1; (void) hipEventCreate(&e0); (void) hipEventCreate(&e1);
		std::vector<float> times;
		for (int rep = 0; rep < 9; rep++) {
			(void) hipEventRecord(e0, 0);
			copy_chunks<D><<<grid, 256>>>(out, in, rows, rpc);
			(void) hipEventRecord(e1, 0); (void) hipEventSynchronize(e1);
			float ms = 0; (void) hipEventElapsedTime(&ms, e0, e1); times.push_back(ms);
		}
		std::sort(times.begin(), times.end());
		const double bytes = (double) rows * 1024 * (1.0 + 1.0 / D);
		std::printf("%s chunk %3u rows (%3u KiB): %7.1f us  %5.1f %% of 8 TB/s\n", what, rpc, rpc, times[4] * 1e3, bytes / (times[4] * 1e-3) / 8e12 * 100);
	}
}

int main(int argc, char** argv)
{
	const uint64_t rows = (uint64_t) 1 << (argc > 1 ? atoi(argv[1]) : 20);   // 2^20 rows: 1 GiB of output (2^22: the headline column's 4 GiB — beyond what the Infinity Cache keeps of the input between repetitions)
	v4 *out, *in;
	(void) hipMalloc(&out, rows * 1024); (void) hipMalloc(&in, rows * 1024);
	(void) hipMemset(in, 1, rows * 1024);
	run<1>(out, in, rows, "copy 1:1 ");
	run<2>(out, in, rows, "input 1/2");
	run<4>(out, in, rows, "input 1/4");
	return 0;
}
