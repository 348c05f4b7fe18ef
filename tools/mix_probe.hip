// What does a READ stream cost a store stream?  (round-4 verdict, next 5.)  Model — stores only — decodes at 85 % of the 8 TB/s HBM peak, Dictionary
// (1 byte read per 4 written) and Frame of Reference (2 per 4) at 79 %, the copy (4 per 4) at 79 %.  Synthetic kernels in the decoders' own shape
// — 256 threads, a tile per workgroup in address order, K 16-byte `sc1 nt` stores per lane per tile, all of a tile's loads before its first store —
// with R bytes read per 16 bytes written, at the residencies the decoders use:
//     hipcc -O3 --offload-arch=gfx950 tools/mix_probe.hip -o tools/mix_probe.bin && tools/mix_probe.bin [log2 of the output's 16-byte groups = 28]
// Prints, per (K, R, resident workgroups per CU, store policy): ms, GB/s of written + read bytes, % of 8 TB/s.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int Policy>   // 1: nt, 2: sc1 nt
__device__ __forceinline__ void store16(u32x4* p, u32x4 v)
{
	if (Policy == 2) { asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }
	else { __builtin_nontemporal_store(v, p); }
}

// R: bytes read per 16-byte group written (0, 2, 4, 8, 16); the read value is folded into what is stored
template <int K, int R, int Policy>
__global__ void stream(u32x4* __restrict__ out, const uint8_t* __restrict__ in, uint64_t groups)
{
	extern __shared__ uint8_t pad[];   // (dynamic LDS only holds the residency down)
	const uint64_t per_tile = (uint64_t) blockDim.x * K;
	for (uint64_t tile = blockIdx.x; tile * per_tile < groups; tile += gridDim.x) {
		const uint64_t g0 = tile * per_tile + threadIdx.x;
		uint32_t got[K];
		#pragma unroll
		for (int k = 0; k < K; k++) {
			const uint64_t g = g0 + (uint64_t) k * blockDim.x;
			got[k] = 0;
			if (g >= groups) { continue; }
			if (R == 2) { got[k] = reinterpret_cast<const uint16_t*>(in)[g]; }
			if (R == 4) { got[k] = reinterpret_cast<const uint32_t*>(in)[g]; }
			if (R == 8) { const u32x2 v = reinterpret_cast<const u32x2*>(in)[g]; got[k] = v.x ^ v.y; }
			if (R == 16) { const u32x4 v = reinterpret_cast<const u32x4*>(in)[g]; got[k] = v.x ^ v.y ^ v.z ^ v.w; }
		}
		#pragma unroll
		for (int k = 0; k < K; k++) {
			const uint64_t g = g0 + (uint64_t) k * blockDim.x;
			if (g >= groups) { continue; }
			const uint32_t base = (uint32_t) g * 4u + got[k];
			store16<Policy>(out + g, u32x4 { base, base + 1, base + 2, base + 3 });
		}
	}
}

// The same stream by a PERSISTENT grid: workgroup b takes tiles b, b + grid, ... and requests tile t + grid's input before it stores tile t — the reads run a whole
// sweep ahead of the stores, so the stores' window can be as narrow as K allows without the loads' latency being exposed (if the workgroups keep pace)
template <int K, int R, int Policy>
__global__ void stream_persistent(u32x4* __restrict__ out, const uint8_t* __restrict__ in, uint64_t groups)
{
	extern __shared__ uint8_t pad[];
	const uint64_t per_tile = (uint64_t) blockDim.x * K;
	uint32_t got[K], next[K];
	auto load = [&](uint64_t tile, uint32_t (&into)[K]) {
		#pragma unroll
		for (int k = 0; k < K; k++) {
			const uint64_t g = tile * per_tile + threadIdx.x + (uint64_t) k * blockDim.x;
			into[k] = 0;
			if (g >= groups) { continue; }
			if (R == 4) { into[k] = reinterpret_cast<const uint32_t*>(in)[g]; }
			if (R == 8) { const u32x2 v = reinterpret_cast<const u32x2*>(in)[g]; into[k] = v.x ^ v.y; }
			if (R == 16) { const u32x4 v = reinterpret_cast<const u32x4*>(in)[g]; into[k] = v.x ^ v.y ^ v.z ^ v.w; }
		}
	};
	load(blockIdx.x, got);
	for (uint64_t tile = blockIdx.x; tile * per_tile < groups; tile += gridDim.x) {
		load(tile + gridDim.x, next);
		#pragma unroll
		for (int k = 0; k < K; k++) {
			const uint64_t g = tile * per_tile + threadIdx.x + (uint64_t) k * blockDim.x;
			if (g >= groups) { continue; }
			const uint32_t base = (uint32_t) g * 4u + got[k];
			store16<Policy>(out + g, u32x4 { base, base + 1, base + 2, base + 3 });
		}
		#pragma unroll
		for (int k = 0; k < K; k++) { got[k] = next[k]; }
	}
}

// R bytes per group read and nothing written but one dword per workgroup (so that the loads cannot be dropped)
template <int K, int R>
__global__ void read_only(uint32_t* __restrict__ sink, const uint8_t* __restrict__ in, uint64_t groups)
{
	const uint64_t per_tile = (uint64_t) blockDim.x * K;
	uint32_t acc = 0;
	for (uint64_t tile = blockIdx.x; tile * per_tile < groups; tile += gridDim.x) {
		const uint64_t g0 = tile * per_tile + threadIdx.x;
		#pragma unroll
		for (int k = 0; k < K; k++) {
			const uint64_t g = g0 + (uint64_t) k * blockDim.x;
			if (g >= groups) { continue; }
			if (R == 4) { acc ^= reinterpret_cast<const uint32_t*>(in)[g]; }
			if (R == 16) { const u32x4 v = reinterpret_cast<const u32x4*>(in)[g]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
		}
	}
	if (acc == 0x12345678u) { sink[blockIdx.x] = acc; }
}

template <typename F>
double time_ms(F f, int reps = 9)
{
	hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
	f(); f();
	std::vector<float> t;
	for (int i = 0; i < reps; i++) { CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b)); float ms; CHECK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
	std::sort(t.begin(), t.end());
	CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
	return t[t.size() / 2];
}

// dynamic LDS that lets at most `resident` workgroups onto a CU of 160 KiB (0: none)
static unsigned lds_for(int resident) { return resident > 0 ? (unsigned) ((160 * 1024 - 1024) / resident / 256 * 256) : 0u; }

template <int K, int R, int Policy>
void line(u32x4* out, const uint8_t* in, uint64_t groups, int resident)
{
	const uint64_t tiles = (groups + 256ull * K - 1) / (256ull * K);
	const unsigned lds = lds_for(resident);
	if (lds > 65536) { CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(stream<K, R, Policy>), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds)); }
	const double ms = time_ms([&] { stream<K, R, Policy><<<(unsigned) tiles, 256, lds>>>(out, in, groups); });
	const double bytes = (16.0 + R) * groups;
	printf("stores %d per lane per tile, %2d bytes read per 16 written, %s, %s workgroups per CU: %7.3f ms  %7.1f GB/s = %4.1f %% of 8 TB/s  (written alone: %4.1f %%)\n",
		K, R, Policy == 2 ? "sc1 nt" : "nt    ", resident ? std::to_string(resident).c_str() : "all", ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80, 16.0 * groups / ms / 1e6 / 80);
	fflush(stdout);
}

template <int K, int R, int Policy>
void persistent_line(u32x4* out, const uint8_t* in, uint64_t groups, int per_cu)
{
	const double ms = time_ms([&] { stream_persistent<K, R, Policy><<<256 * per_cu, 256, 0>>>(out, in, groups); });
	const double bytes = (16.0 + R) * groups;
	printf("PERSISTENT grid of %d workgroups per CU, reads a sweep ahead: stores %d per lane per tile, %2d bytes read per 16 written, %s: %7.3f ms  %7.1f GB/s = %4.1f %% of 8 TB/s\n",
		per_cu, K, R, Policy == 2 ? "sc1 nt" : "nt    ", ms, bytes / ms / 1e6, bytes / ms / 1e6 / 80);
	fflush(stdout);
}

template <int K, int R>
void lines(u32x4* out, const uint8_t* in, uint64_t groups)
{
	for (int resident : { 0, 6, 4, 3 }) { line<K, R, 2>(out, in, groups, resident); }
	line<K, R, 1>(out, in, groups, 0);
	line<K, R, 1>(out, in, groups, 6);
}

int main(int argc, char** argv)
{
	const int log2_groups = argc > 1 ? atoi(argv[1]) : 28;
	const uint64_t groups = 1ull << log2_groups;
	u32x4* out; uint8_t* in; uint32_t* sink;
	CHECK(hipMalloc(&out, groups * 16)); CHECK(hipMalloc(&in, groups * 16)); CHECK(hipMalloc(&sink, 1 << 24));
	CHECK(hipMemset(in, 1, groups * 16));
	printf("# %llu groups of 16 bytes written (%.2f GB), one tile per workgroup of 256 threads\n", (unsigned long long) groups, groups * 16 / 1e9);
	lines<1, 0>(out, in, groups);    // Model's shape
	lines<4, 0>(out, in, groups);
	lines<4, 4>(out, in, groups);    // Dictionary <4,4,1>: a dword of four keys per store
	lines<2, 8>(out, in, groups);    // Frame of Reference <4,int,short>: 8 bytes of four offsets per store, two stores per tile
	lines<4, 8>(out, in, groups);
	lines<4, 16>(out, in, groups);   // the copy
	lines<4, 2>(out, in, groups);
	lines<1, 4>(out, in, groups);    // ... with tiles of ONE store per lane, as Model's
	lines<2, 4>(out, in, groups);
	lines<1, 8>(out, in, groups);
	lines<1, 16>(out, in, groups);
	lines<8, 4>(out, in, groups);
	for (int per_cu : { 4, 6, 8 }) {
		persistent_line<1, 4, 2>(out, in, groups, per_cu); persistent_line<2, 4, 2>(out, in, groups, per_cu); persistent_line<4, 4, 2>(out, in, groups, per_cu);
		persistent_line<1, 8, 2>(out, in, groups, per_cu); persistent_line<2, 8, 2>(out, in, groups, per_cu);
		persistent_line<1, 16, 2>(out, in, groups, per_cu); persistent_line<4, 16, 2>(out, in, groups, per_cu);
	}
	for (int resident : { 0 }) {
		(void) resident;
		const uint64_t tiles = (groups + 1023) / 1024;
		double ms = time_ms([&] { read_only<4, 16><<<(unsigned) tiles, 256>>>(sink, in, groups); });
		printf("reads only, 16 bytes per lane x 4 per tile: %7.3f ms %7.1f GB/s = %4.1f %% of 8 TB/s\n", ms, 16.0 * groups / ms / 1e6, 16.0 * groups / ms / 1e6 / 80);
		ms = time_ms([&] { read_only<4, 4><<<(unsigned) tiles, 256>>>(sink, in, groups); });
		printf("reads only,  4 bytes per lane x 4 per tile: %7.3f ms %7.1f GB/s = %4.1f %% of 8 TB/s\n", ms, 4.0 * groups / ms / 1e6, 4.0 * groups / ms / 1e6 / 80);
	}
	return 0;
}
