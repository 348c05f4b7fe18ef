// Development probe: what HBM rate does MI355X sustain for streams shaped like the decoders'
// (1 byte read : 4 bytes written etc.), as a function of grid size, store flavour and unroll?
// hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o /tmp/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while(0)

// widen: read 4 B per lane (4 keys), write 16 B per lane. K steps per tile.
template <int K, bool NT>
__global__ void widen_1_4(u32x4* __restrict__ out, const uint32_t* __restrict__ in, uint64_t groups)
{
	const uint64_t per_tile = (uint64_t) blockDim.x * K;
	for (uint64_t tile = blockIdx.x; tile * per_tile < groups; tile += gridDim.x) {
		uint64_t g0 = tile * per_tile + threadIdx.x;
		uint32_t v[K];
		#pragma unroll
		for (int k = 0; k < K; k++) { uint64_t g = g0 + (uint64_t) k * blockDim.x; if (g < groups) v[k] = in[g]; }
		#pragma unroll
		for (int k = 0; k < K; k++) {
			uint64_t g = g0 + (uint64_t) k * blockDim.x; if (g >= groups) continue;
			u32x4 o = { v[k] & 0xff, (v[k] >> 8) & 0xff, (v[k] >> 16) & 0xff, v[k] >> 24 };
			if (NT) __builtin_nontemporal_store(o, out + g); else out[g] = o;
		}
	}
}

template <int K, bool NT>
__global__ void fill(u32x4* __restrict__ out, uint64_t groups)
{
	const uint64_t per_tile = (uint64_t) blockDim.x * K;
	for (uint64_t tile = blockIdx.x; tile * per_tile < groups; tile += gridDim.x) {
		uint64_t g0 = tile * per_tile + threadIdx.x;
		#pragma unroll
		for (int k = 0; k < K; k++) {
			uint64_t g = g0 + (uint64_t) k * blockDim.x; if (g >= groups) continue;
			u32x4 o = { (uint32_t) g, 1, 2, 3 };
			if (NT) __builtin_nontemporal_store(o, out + g); else out[g] = o;
		}
	}
}

template <int K, bool NT>
__global__ void copy(u32x4* __restrict__ out, const u32x4* __restrict__ in, uint64_t groups)
{
	const uint64_t per_tile = (uint64_t) blockDim.x * K;
	for (uint64_t tile = blockIdx.x; tile * per_tile < groups; tile += gridDim.x) {
		uint64_t g0 = tile * per_tile + threadIdx.x;
		u32x4 v[K];
		#pragma unroll
		for (int k = 0; k < K; k++) { uint64_t g = g0 + (uint64_t) k * blockDim.x; if (g < groups) v[k] = NT ? __builtin_nontemporal_load(in + g) : in[g]; }
		#pragma unroll
		for (int k = 0; k < K; k++) { uint64_t g = g0 + (uint64_t) k * blockDim.x; if (g < groups) { if (NT) __builtin_nontemporal_store(v[k], out + g); else out[g] = v[k]; } }
	}
}

template <int K>
__global__ void readonly(uint32_t* __restrict__ sink, const u32x4* __restrict__ in, uint64_t groups)
{
	const uint64_t per_tile = (uint64_t) blockDim.x * K;
	uint32_t acc = 0;
	for (uint64_t tile = blockIdx.x; tile * per_tile < groups; tile += gridDim.x) {
		uint64_t g0 = tile * per_tile + threadIdx.x;
		u32x4 v[K];
		#pragma unroll
		for (int k = 0; k < K; k++) { uint64_t g = g0 + (uint64_t) k * blockDim.x; if (g < groups) v[k] = in[g]; else v[k] = u32x4{0,0,0,0}; }
		#pragma unroll
		for (int k = 0; k < K; k++) acc += v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
	}
	if (acc == 0x12345678) sink[0] = acc;
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
	const uint64_t n = 1ull << 30;          // elements
	u32x4* out; uint32_t* in; u32x4* in16;
	CHECK(hipMalloc(&out, n * 4)); CHECK(hipMalloc(&in, n)); CHECK(hipMalloc(&in16, n * 4));
	CHECK(hipMemset(in, 1, n)); CHECK(hipMemset(in16, 1, n * 4));
	const uint64_t groups = n / 4;
	int grids[] = { 1024, 2048, 4096, 8192, 16384, 65536, 262144 };
	for (int block : { 256, 512, 1024 }) {
		for (int grid : grids) {
			double t;
			t = time_ms([&] { widen_1_4<4, true><<<grid, block>>>(out, in, groups); });
			printf("widen1:4 K4 nt   block %4d grid %6d: %7.3f ms %7.1f GB/s\n", block, grid, t, 5.0 * n / t / 1e6);
			t = time_ms([&] { widen_1_4<4, false><<<grid, block>>>(out, in, groups); });
			printf("widen1:4 K4 plain block %4d grid %6d: %7.3f ms %7.1f GB/s\n", block, grid, t, 5.0 * n / t / 1e6);
		}
	}
	for (int grid : grids) {
		double t;
		t = time_ms([&] { widen_1_4<8, true><<<grid, 256>>>(out, in, groups); });
		printf("widen1:4 K8 nt   block  256 grid %6d: %7.3f ms %7.1f GB/s\n", grid, t, 5.0 * n / t / 1e6);
		t = time_ms([&] { widen_1_4<1, true><<<grid, 256>>>(out, in, groups); });
		printf("widen1:4 K1 nt   block  256 grid %6d: %7.3f ms %7.1f GB/s\n", grid, t, 5.0 * n / t / 1e6);
		t = time_ms([&] { widen_1_4<2, true><<<grid, 256>>>(out, in, groups); });
		printf("widen1:4 K2 nt   block  256 grid %6d: %7.3f ms %7.1f GB/s\n", grid, t, 5.0 * n / t / 1e6);
		t = time_ms([&] { fill<4, true><<<grid, 256>>>(out, groups); });
		printf("fill     K4 nt   block  256 grid %6d: %7.3f ms %7.1f GB/s\n", grid, t, 4.0 * n / t / 1e6);
		t = time_ms([&] { fill<4, false><<<grid, 256>>>(out, groups); });
		printf("fill     K4 plain block  256 grid %6d: %7.3f ms %7.1f GB/s\n", grid, t, 4.0 * n / t / 1e6);
		t = time_ms([&] { copy<4, true><<<grid, 256>>>(out, in16, groups); });
		printf("copy     K4 nt   block  256 grid %6d: %7.3f ms %7.1f GB/s\n", grid, t, 8.0 * n / t / 1e6);
		t = time_ms([&] { copy<4, false><<<grid, 256>>>(out, in16, groups); });
		printf("copy     K4 plain block  256 grid %6d: %7.3f ms %7.1f GB/s\n", grid, t, 8.0 * n / t / 1e6);
		t = time_ms([&] { readonly<4><<<grid, 256>>>((uint32_t*) out, in16, groups); });
		printf("read     K4       block  256 grid %6d: %7.3f ms %7.1f GB/s\n", grid, t, 4.0 * n / t / 1e6);
	}
	{ double t = time_ms([&] { CHECK(hipMemsetAsync(out, 0, n * 4)); }); printf("hipMemset 4GiB: %7.3f ms %7.1f GB/s\n", t, 4.0 * n / t / 1e6); }
	{ double t = time_ms([&] { CHECK(hipMemcpyAsync(out, in16, n * 4, hipMemcpyDeviceToDevice)); }); printf("hipMemcpy D2D 4GiB: %7.3f ms %7.1f GB/s (r+w)\n", t, 8.0 * n / t / 1e6); }
	return 0;
}
