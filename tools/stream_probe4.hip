// Development probe 4: u8 -> u64 widening stream (1 B read : 8 B written, the shape of Dictionary / DZB-fixed / Delta
// with 1-byte inputs and 8-byte outputs) — are the 2-byte-per-lane loads the problem, and does a wide load
// redistributed through wave-private LDS fix it?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while(0)

__device__ inline u32x4 widen(uint32_t two_bytes) { return u32x4{ two_bytes & 0xff, 0, (two_bytes >> 8) & 0xff, 0 }; }

// (a) tile per block, K steps, each lane loads 2 bytes per step (today's streaming decoders)
template <int K>
__global__ void narrow_loads(u32x4* __restrict__ out, const uint16_t* __restrict__ in, uint64_t groups)
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

// (b) each wavefront takes K consecutive rows (row = 64 groups = 128 bytes in, 1 KiB out): the K*128 input bytes are
//     loaded as 16 bytes per lane by the first K*8 lanes, passed through wave-private LDS, and read back 2 bytes per lane per row
template <int K>
__global__ void wide_loads_through_lds(u32x4* __restrict__ out, const uint8_t* __restrict__ in, uint64_t groups)
{
	__shared__ __attribute__((aligned(16))) uint8_t staging[16][K * 128];
	const unsigned lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint64_t waves_per_block = blockDim.x >> 6;
	const uint64_t rows = groups / 64;
	for (uint64_t first_row = ((uint64_t) blockIdx.x * waves_per_block + w) * K; first_row < rows; first_row += (uint64_t) gridDim.x * waves_per_block * K) {
		if (lane < K * 8) {
			u32x4 piece = *reinterpret_cast<const u32x4*>(in + first_row * 128 + lane * 16);
			*reinterpret_cast<u32x4*>(&staging[w][lane * 16]) = piece;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		uint32_t v[K];
		#pragma unroll
		for (int k = 0; k < K; k++) v[k] = *reinterpret_cast<const uint16_t*>(&staging[w][k * 128 + lane * 2]);
		#pragma unroll
		for (int k = 0; k < K; k++) __builtin_nontemporal_store(widen(v[k]), out + (first_row + k) * 64 + lane);
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	}
}

// (c) as (a) but 4-byte loads: every lane loads 4 bytes = two groups, and stores two 16-byte pieces 32 bytes apart... no:
//     lane l of step k takes groups 2l and 2l+1 of a double row: two stores, each contiguous over HALF the lanes' pieces
template <int K>
__global__ void dword_loads_two_stores(u32x4* __restrict__ out, const uint32_t* __restrict__ in, uint64_t groups)
{
	const uint64_t pairs = groups / 2, per_tile = (uint64_t) blockDim.x * K;
	for (uint64_t tile = blockIdx.x; tile * per_tile < pairs; tile += gridDim.x) {
		uint64_t p0 = tile * per_tile + threadIdx.x;
		uint32_t v[K];
		#pragma unroll
		for (int k = 0; k < K; k++) v[k] = in[p0 + (uint64_t) k * blockDim.x];
		#pragma unroll
		for (int k = 0; k < K; k++) {
			uint64_t p = p0 + (uint64_t) k * blockDim.x;
			__builtin_nontemporal_store(widen(v[k] & 0xffff), out + 2 * p);
			__builtin_nontemporal_store(widen(v[k] >> 16), out + 2 * p + 1);
		}
	}
}

// (d) the Delta wave path's loop without the scan: a wavefront owns 32 consecutive rows, R rows in flight, the next
//     group's 2-byte loads issued before this group's stores
template <int R>
__global__ void wave_chunks_narrow(u32x4* __restrict__ out, const uint16_t* __restrict__ in, uint64_t groups)
{
	const unsigned lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint64_t rows = groups / 64, waves_per_block = blockDim.x >> 6;
	for (uint64_t first_row = ((uint64_t) blockIdx.x * waves_per_block + w) * 32; first_row < rows; first_row += (uint64_t) gridDim.x * waves_per_block * 32) {
		uint32_t v[R], next[R];
		#pragma unroll
		for (int k = 0; k < R; k++) v[k] = in[(first_row + k) * 64 + lane];
		for (int r = 0; r < 32; r += R) {
			if (r + R < 32) {
				#pragma unroll
				for (int k = 0; k < R; k++) next[k] = in[(first_row + r + R + k) * 64 + lane];
			}
			#pragma unroll
			for (int k = 0; k < R; k++) __builtin_nontemporal_store(widen(v[k]), out + (first_row + r + k) * 64 + lane);
			#pragma unroll
			for (int k = 0; k < R; k++) v[k] = next[k];
		}
	}
}

// (e) the same chunk of 32 rows, but the input of K rows at a time arrives as one 16-byte-per-lane load by the first K*8 lanes
//     and is redistributed through wave-private LDS; the next group's load is issued before this group's stores
template <int K>
__global__ void wave_chunks_wide(u32x4* __restrict__ out, const uint8_t* __restrict__ in, uint64_t groups)
{
	__shared__ __attribute__((aligned(16))) uint8_t staging[4][2][K * 128];
	const unsigned lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint64_t rows = groups / 64, waves_per_block = blockDim.x >> 6;
	for (uint64_t first_row = ((uint64_t) blockIdx.x * waves_per_block + w) * 32; first_row < rows; first_row += (uint64_t) gridDim.x * waves_per_block * 32) {
		u32x4 piece = u32x4{0, 0, 0, 0}, next_piece = u32x4{0, 0, 0, 0};
		if (lane < K * 8) piece = *reinterpret_cast<const u32x4*>(in + first_row * 128 + lane * 16);
		int buffer = 0;
		for (int r = 0; r < 32; r += K) {
			if (r + K < 32 && lane < K * 8) next_piece = *reinterpret_cast<const u32x4*>(in + (first_row + r + K) * 128 + lane * 16);
			if (lane < K * 8) *reinterpret_cast<u32x4*>(&staging[w][buffer][lane * 16]) = piece;
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			#pragma unroll
			for (int k = 0; k < K; k++) {
				uint32_t v = *reinterpret_cast<const uint16_t*>(&staging[w][buffer][k * 128 + lane * 2]);
				__builtin_nontemporal_store(widen(v), out + (first_row + r + k) * 64 + lane);
			}
			piece = next_piece; buffer ^= 1;
		}
	}
}

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
	const uint64_t n = 1ull << 30;   // elements: 1 GiB in, 8 GiB out
	u32x4* out; uint8_t* in;
	CHECK(hipMalloc(&out, n * 8)); CHECK(hipMalloc(&in, n)); CHECK(hipMemset(in, 1, n));
	const uint64_t groups = n / 2;
	const double bytes = 9.0 * n;
	auto report = [&](const char* name, double ms) { printf("%-60s %8.3f ms  %7.1f GB/s (%.1f%% of 8 TB/s)\n", name, ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80); };
	report("narrow 2-byte loads, tile of 4 steps x 256 threads", time_ms([&] { narrow_loads<4><<<groups / (256 * 4), 256>>>(out, (const uint16_t*) in, groups); }));
	report("narrow 2-byte loads, tile of 8 steps x 256 threads", time_ms([&] { narrow_loads<8><<<groups / (256 * 8), 256>>>(out, (const uint16_t*) in, groups); }));
	report("wide loads through LDS, 4 rows per wavefront", time_ms([&] { wide_loads_through_lds<4><<<groups / 64 / (4 * 4), 256>>>(out, in, groups); }));
	report("wide loads through LDS, 8 rows per wavefront", time_ms([&] { wide_loads_through_lds<8><<<groups / 64 / (4 * 8), 256>>>(out, in, groups); }));
	report("4-byte loads, two stores per lane", time_ms([&] { dword_loads_two_stores<4><<<groups / 2 / (256 * 4), 256>>>(out, (const uint32_t*) in, groups); }));
	report("4-byte loads, two stores per lane, 2 steps", time_ms([&] { dword_loads_two_stores<2><<<groups / 2 / (256 * 2), 256>>>(out, (const uint32_t*) in, groups); }));
	const unsigned chunk_grid = (unsigned) (groups / 64 / 32 / 4);
	report("wave chunks of 32 rows, 2-byte loads, 2 rows in flight", time_ms([&] { wave_chunks_narrow<2><<<chunk_grid, 256>>>(out, (const uint16_t*) in, groups); }));
	report("wave chunks of 32 rows, 2-byte loads, 4 rows in flight", time_ms([&] { wave_chunks_narrow<4><<<chunk_grid, 256>>>(out, (const uint16_t*) in, groups); }));
	report("wave chunks of 32 rows, 2-byte loads, 8 rows in flight", time_ms([&] { wave_chunks_narrow<8><<<chunk_grid, 256>>>(out, (const uint16_t*) in, groups); }));
	report("wave chunks of 32 rows, wide loads via LDS, 4 rows a group", time_ms([&] { wave_chunks_wide<4><<<chunk_grid, 256>>>(out, in, groups); }));
	report("wave chunks of 32 rows, wide loads via LDS, 8 rows a group", time_ms([&] { wave_chunks_wide<8><<<chunk_grid, 256>>>(out, in, groups); }));
	CHECK(hipDeviceSynchronize());
	return 0;
}
