// Development probe 2: dictionary-shaped stream (u8 keys -> u32 via LDS), one tile per block vs
// grid stride, tile size sweep.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while(0)

template <int K, bool NT, bool LDS>
__global__ void dict(u32x4* __restrict__ out, const uint32_t* __restrict__ in, const uint32_t* __restrict__ d, uint64_t groups)
{
	__shared__ uint32_t s[256];
	if (LDS) { for (int i = threadIdx.x; i < 256; i += blockDim.x) s[i] = d[i]; __syncthreads(); }
	const uint32_t* t = LDS ? s : d;
	const uint64_t per_tile = (uint64_t) blockDim.x * K;
	for (uint64_t tile = blockIdx.x; tile * per_tile < groups; tile += gridDim.x) {
		uint64_t g0 = tile * per_tile + threadIdx.x;
		uint32_t v[K];
		#pragma unroll
		for (int k = 0; k < K; k++) { uint64_t g = g0 + (uint64_t) k * blockDim.x; if (g < groups) v[k] = in[g]; }
		#pragma unroll
		for (int k = 0; k < K; k++) {
			uint64_t g = g0 + (uint64_t) k * blockDim.x; if (g >= groups) continue;
			u32x4 o = { t[v[k] & 0xff], t[(v[k] >> 8) & 0xff], t[(v[k] >> 16) & 0xff], t[v[k] >> 24] };
			if (NT) __builtin_nontemporal_store(o, out + g); else out[g] = o;
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

template <int K, bool NT, bool LDS>
void sweep(u32x4* out, const uint32_t* in, const uint32_t* d, uint64_t groups, uint64_t n, int block)
{
	uint64_t tiles = (groups + (uint64_t) block * K - 1) / ((uint64_t) block * K);
	for (uint64_t grid : { (uint64_t) 2048, (uint64_t) 8192, tiles / 4, tiles }) {
		if (grid == 0 || grid > tiles) continue;
		double t = time_ms([&] { dict<K, NT, LDS><<<(unsigned) grid, block>>>(out, in, d, groups); });
		printf("dict K%-2d %s %s block %4d grid %8llu (tiles %8llu): %7.3f ms %7.1f GB/s\n", K, NT ? "nt   " : "plain", LDS ? "lds" : "glb", block, (unsigned long long) grid, (unsigned long long) tiles, t, 5.0 * n / t / 1e6);
	}
}

int main()
{
	const uint64_t n = 1ull << 30;
	u32x4* out; uint32_t* in; uint32_t* d;
	CHECK(hipMalloc(&out, n * 4)); CHECK(hipMalloc(&in, n)); CHECK(hipMalloc(&d, 1024));
	std::vector<uint32_t> h(n / 4); uint64_t x = 88172645463325252ull;
	for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (uint32_t) x; }
	CHECK(hipMemcpy(in, h.data(), n, hipMemcpyHostToDevice)); CHECK(hipMemset(d, 3, 1024));
	const uint64_t groups = n / 4;
	for (int block : { 64, 128, 256, 512 }) {
		sweep<1, true, true>(out, in, d, groups, n, block);
		sweep<2, true, true>(out, in, d, groups, n, block);
		sweep<4, true, true>(out, in, d, groups, n, block);
		sweep<8, true, true>(out, in, d, groups, n, block);
		sweep<16, true, true>(out, in, d, groups, n, block);
		sweep<4, false, true>(out, in, d, groups, n, block);
		sweep<8, false, true>(out, in, d, groups, n, block);
	}
	sweep<2, true, false>(out, in, d, groups, n, 256);
	sweep<4, true, false>(out, in, d, groups, n, 256);
	return 0;
}
