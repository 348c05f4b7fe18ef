// Known-traffic kernels for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// (MI355X_MICROARCH.md §HBM: FETCH_SIZE reads exactly half of a wide coalesced stream's bytes; other
// widths and WRITE_SIZE must be calibrated on a known byte count in one's own access pattern).
// Each kernel moves exactly 1 GiB (reads) or 1 GiB (writes) with the load/store width our decoders use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename T>
__global__ void calibrate_read(uint32_t* sink, const T* __restrict__ in, uint64_t count)
{
	uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	uint32_t acc = 0;
	for (int k = 0; k < 4; k++) { uint64_t j = i + (uint64_t) k * gridDim.x * blockDim.x; if (j < count) { T v = in[j]; acc ^= *reinterpret_cast<uint32_t*>(&v); } }
	if (acc == 0x12345679u) { sink[0] = acc; }
}
template <typename T>
__global__ void calibrate_write(T* __restrict__ out, uint64_t count)
{
	uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	for (int k = 0; k < 4; k++) { uint64_t j = i + (uint64_t) k * gridDim.x * blockDim.x; if (j < count) { T v; __builtin_memset(&v, 0x5a, sizeof(T)); __builtin_nontemporal_store(v, out + j); } }
}

int main()
{
	const uint64_t bytes = 1ull << 30;
	void *a, *b;
	CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes));
	CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 2, bytes));
	CHECK(hipDeviceSynchronize());
	auto grid = [&](uint64_t count) { return (unsigned) ((count + 4 * 256 - 1) / (4 * 256)); };
	calibrate_read<uint32_t><<<grid(bytes / 4), 256>>>((uint32_t*) b, (const uint32_t*) a, bytes / 4);
	calibrate_read<u32x2><<<grid(bytes / 8), 256>>>((uint32_t*) b, (const u32x2*) a, bytes / 8);
	calibrate_read<u32x4><<<grid(bytes / 16), 256>>>((uint32_t*) b, (const u32x4*) a, bytes / 16);
	calibrate_write<u32x4><<<grid(bytes / 16), 256>>>((u32x4*) b, bytes / 16);
	calibrate_write<uint32_t><<<grid(bytes / 4), 256>>>((uint32_t*) b, bytes / 4);
	CHECK(hipDeviceSynchronize());
	printf("calibration kernels done: each moved %llu bytes\n", (unsigned long long) bytes);
	return 0;
}
