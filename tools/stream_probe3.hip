// Development probe 3: delta-shaped stream (u16 -> u64, 2 elements per lane per row of 128) under different
// work-to-wavefront mappings, WITHOUT the scan — isolates the memory-access-pattern cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while(0)

__device__ inline u32x4 widen(uint32_t v) { return u32x4{ v & 0xffff, 0, v >> 16, 0 }; }

// (a) wave per segment of SEG rows, R rows in flight
template <int R>
__global__ void wave_per_segment(u32x4* __restrict__ out, const uint32_t* __restrict__ in, uint64_t rows, int seg_rows)
{
	const unsigned lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint64_t wave = (uint64_t) blockIdx.x * (blockDim.x >> 6) + w, nw = (uint64_t) gridDim.x * (blockDim.x >> 6);
	for (uint64_t s = wave; s * seg_rows < rows; s += nw) {
		for (int r0 = 0; r0 < seg_rows; r0 += R) {
			uint32_t v[R];
			#pragma unroll
			for (int k = 0; k < R; k++) v[k] = in[(s * seg_rows + r0 + k) * 64 + lane];
			#pragma unroll
			for (int k = 0; k < R; k++) __builtin_nontemporal_store(widen(v[k]), out + (s * seg_rows + r0 + k) * 64 + lane);
		}
	}
}
// (b) tile per block: K steps, rows interleaved across the block's waves (dictionary-like)
template <int K>
__global__ void tile_per_block(u32x4* __restrict__ out, const uint32_t* __restrict__ in, uint64_t groups)
{
	const uint64_t per_tile = (uint64_t) blockDim.x * K;
	for (uint64_t tile = blockIdx.x; tile * per_tile < groups; tile += gridDim.x) {
		uint64_t g0 = tile * per_tile + threadIdx.x;
		uint32_t v[K];
		#pragma unroll
		for (int k = 0; k < K; k++) v[k] = in[g0 + (uint64_t) k * blockDim.x];
		#pragma unroll
		for (int k = 0; k < K; k++) __builtin_nontemporal_store(widen(v[k]), out + g0 + (uint64_t) k * blockDim.x);
	}
}
// (c) wave per segment but each wave's rows of a block-tile are adjacent: block handles 4 consecutive segments; same as (a).
// (d) wave per segment, all rows loaded first (R = seg_rows = 32), then stored
template <typename F>
double time_ms(F f, int reps = 10)
{
	hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
	f(); f();
	std::vector<float> t;
	for (int i = 0; i < reps; i++) { CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
	std::sort(t.begin(), t.end());
	return t[t.size() / 2];
}
int main()
{
	const uint64_t n = 1ull << 30;   // elements: 2 GiB in, 8 GiB out
	u32x4* out; uint32_t* in;
	CHECK(hipMalloc(&out, n * 8)); CHECK(hipMalloc(&in, n * 2)); CHECK(hipMemset(in, 1, n * 2));
	const uint64_t groups = n / 2, rows = groups / 64;
	const double bytes = 10.0 * n;
	for (int rep = 0; rep < 2; rep++) {
	for (int block : { 64, 128, 256 }) {
		unsigned grid = (unsigned) (rows / 32 / (block / 64));
		double t;
		t = time_ms([&] { wave_per_segment<1><<<grid, block>>>(out, in, rows, 32); }); printf("wave/segment R1  block %3d: %.3f ms %7.1f GB/s\n", block, t, bytes / t / 1e6);
		t = time_ms([&] { wave_per_segment<2><<<grid, block>>>(out, in, rows, 32); }); printf("wave/segment R2  block %3d: %.3f ms %7.1f GB/s\n", block, t, bytes / t / 1e6);
		t = time_ms([&] { wave_per_segment<4><<<grid, block>>>(out, in, rows, 32); }); printf("wave/segment R4  block %3d: %.3f ms %7.1f GB/s\n", block, t, bytes / t / 1e6);
		t = time_ms([&] { wave_per_segment<8><<<grid, block>>>(out, in, rows, 32); }); printf("wave/segment R8  block %3d: %.3f ms %7.1f GB/s\n", block, t, bytes / t / 1e6);
		t = time_ms([&] { wave_per_segment<32><<<grid, block>>>(out, in, rows, 32); }); printf("wave/segment R32 block %3d: %.3f ms %7.1f GB/s\n", block, t, bytes / t / 1e6);
	}
	for (int K : { 2, 4, 8 }) {
		double t;
		if (K == 2) { t = time_ms([&] { tile_per_block<2><<<(unsigned) (groups / (256 * 2)), 256>>>(out, in, groups); }); }
		else if (K == 4) { t = time_ms([&] { tile_per_block<4><<<(unsigned) (groups / (256 * 4)), 256>>>(out, in, groups); }); }
		else { t = time_ms([&] { tile_per_block<8><<<(unsigned) (groups / (256 * 8)), 256>>>(out, in, groups); }); }
		printf("tile/block K%d block 256: %.3f ms %7.1f GB/s\n", K, t, bytes / t / 1e6);
	}
	}
	return 0;
}
