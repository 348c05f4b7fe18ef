// What a READ stream alone gets (development probe, MI355X): workgroups of 256 threads, a wavefront reading consecutive pieces of ROWS rows of 16 bytes per lane
// (all ROWS loads in flight, then reduced), wavefronts over consecutive pieces, workgroups in address order; plain loads and non-temporal ones.
//    hipcc --offload-arch=gfx950 -O3 tools/read_probe.hip -o tools/read_probe.bin && tools/read_probe.bin [log2 bytes = 30]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef uint32_t sixteen_bytes __attribute__((ext_vector_type(4)));

// the cache-policy bits no builtin names, by inline assembly: the loads of a batch, then ONE wait that the data registers pass through
#define PROBE_POLICY_KERNEL(NAME, BITS) \
template <unsigned Rows, unsigned WavesPerSimd> \
__global__ void __launch_bounds__(256, WavesPerSimd) NAME(const sixteen_bytes* __restrict__ in, uint32_t* __restrict__ out, uint64_t pieces_of_16, uint32_t rows_per_wave) \
{ \
	const unsigned lane = threadIdx.x % 64, wave = threadIdx.x / 64; \
	uint32_t sum = 0; \
	const uint64_t wave_first = ((uint64_t) blockIdx.x * 4 + wave) * rows_per_wave * 64; \
	for (uint32_t r0 = 0; r0 < rows_per_wave; r0 += Rows) { \
		sixteen_bytes v[Rows]; \
		_Pragma("unroll") for (unsigned i = 0; i < Rows; i++) { \
			const sixteen_bytes* at = in + wave_first + (uint64_t) (r0 + i) * 64 + lane; \
			asm volatile("global_load_dwordx4 %0, %1, off " BITS : "=v"(v[i]) : "v"(at) : "memory"); \
		} \
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
		_Pragma("unroll") for (unsigned i = 0; i < Rows; i++) { asm volatile("" : "+v"(v[i])); sum += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w; } \
	} \
	if (sum == 0x12345678u) { out[blockIdx.x] = sum; } \
}
PROBE_POLICY_KERNEL(read_rows_nt_asm, "nt")
PROBE_POLICY_KERNEL(read_rows_sc1_nt, "sc1 nt")
PROBE_POLICY_KERNEL(read_rows_sc0_sc1_nt, "sc0 sc1 nt")
PROBE_POLICY_KERNEL(read_rows_sc1, "sc1")
PROBE_POLICY_KERNEL(read_rows_sc0_sc1, "sc0 sc1")
PROBE_POLICY_KERNEL(read_rows_sc0_nt, "sc0 nt")

template <typename Kernel>
static void run_policy(const char* what, Kernel kernel, unsigned rows, unsigned waves, const sixteen_bytes* in, uint32_t* out, uint64_t bytes, uint32_t rows_per_wave)
{
	const uint64_t pieces = bytes / 16, grid = pieces / (4ull * rows_per_wave * 64);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	std::vector<float> times;
	for (int rep = 0; rep < 7; rep++) {
		hipEventRecord(a, 0);
		hipLaunchKernelGGL(kernel, dim3((unsigned) grid), dim3(256), 0, 0, in, out, pieces, rows_per_wave);
		hipEventRecord(b, 0); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b); times.push_back(ms);
	}
	std::sort(times.begin(), times.end());
	printf("%-10s rows in flight %2u, %2u rows per wavefront (%3u KiB per workgroup), %u waves per SIMD: %.3f ms  %7.1f GB/s = %.1f %% of 8 TB/s\n", what, rows, rows_per_wave,
		rows_per_wave * 4, waves, times[3], bytes / times[3] / 1e6, bytes / times[3] / 1e6 / 80);
}

template <unsigned Rows, bool NonTemporal, unsigned WavesPerSimd>
__global__ void __launch_bounds__(256, WavesPerSimd) read_rows(const sixteen_bytes* __restrict__ in, uint32_t* __restrict__ out, uint64_t pieces_of_16, uint32_t rows_per_wave)
{
	const unsigned lane = threadIdx.x % 64, wave = threadIdx.x / 64;
	uint32_t sum = 0;
	const uint64_t wave_first = ((uint64_t) blockIdx.x * 4 + wave) * rows_per_wave * 64;
	for (uint32_t r0 = 0; r0 < rows_per_wave; r0 += Rows) {
		sixteen_bytes v[Rows];
		#pragma unroll
		for (unsigned i = 0; i < Rows; i++) {
			const uint64_t at = wave_first + (uint64_t) (r0 + i) * 64 + lane;
			if (NonTemporal) { v[i] = __builtin_nontemporal_load(in + at); } else { v[i] = in[at]; }
		}
		#pragma unroll
		for (unsigned i = 0; i < Rows; i++) { sum += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w; }
		__builtin_amdgcn_sched_barrier(0);
	}
	if (sum == 0x12345678u) { out[blockIdx.x] = sum; }
}

template <unsigned Rows, bool NonTemporal, unsigned WavesPerSimd>
static void run(const char* what, const sixteen_bytes* in, uint32_t* out, uint64_t bytes, uint32_t rows_per_wave)
{
	const uint64_t pieces = bytes / 16, grid = pieces / (4ull * rows_per_wave * 64);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	std::vector<float> times;
	for (int rep = 0; rep < 7; rep++) {
		hipEventRecord(a, 0);
		hipLaunchKernelGGL((read_rows<Rows, NonTemporal, WavesPerSimd>), dim3((unsigned) grid), dim3(256), 0, 0, in, out, pieces, rows_per_wave);
		hipEventRecord(b, 0); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b); times.push_back(ms);
	}
	std::sort(times.begin(), times.end());
	printf("%-10s rows in flight %2u, %2u rows per wavefront (%3u KiB per workgroup), %u waves per SIMD: %.3f ms  %7.1f GB/s = %.1f %% of 8 TB/s\n", what, Rows, rows_per_wave,
		rows_per_wave * 4, WavesPerSimd, times[3], bytes / times[3] / 1e6, bytes / times[3] / 1e6 / 80);
}

int main(int argc, char** argv)
{
	const uint64_t bytes = 1ull << (argc > 1 ? atoi(argv[1]) : 30);
	sixteen_bytes* in; uint32_t* out;
	hipMalloc(&in, bytes); hipMalloc(&out, 1 << 24);
	hipMemset(in, 1, bytes);
	hipDeviceSynchronize();
	run<4, false, 8>("plain", in, out, bytes, 16);
	run<8, false, 4>("plain", in, out, bytes, 16);
	run<8, false, 8>("plain", in, out, bytes, 16);
	run<16, false, 4>("plain", in, out, bytes, 16);
	run<8, false, 4>("plain", in, out, bytes, 64);
	run<8, false, 8>("plain", in, out, bytes, 8);
	run<4, false, 8>("plain", in, out, bytes, 4);
	run<4, true, 8>("nt", in, out, bytes, 16);
	run<8, true, 4>("nt", in, out, bytes, 16);
	run<8, true, 8>("nt", in, out, bytes, 16);
	run<16, true, 4>("nt", in, out, bytes, 16);
	run<8, true, 8>("nt", in, out, bytes, 8);
	run<4, true, 8>("nt", in, out, bytes, 4);
	run_policy("nt (asm)", read_rows_nt_asm<8, 8>, 8, 8, in, out, bytes, 16);
	run_policy("sc1 nt", read_rows_sc1_nt<8, 8>, 8, 8, in, out, bytes, 16);
	run_policy("sc0 sc1 nt", read_rows_sc0_sc1_nt<8, 8>, 8, 8, in, out, bytes, 16);
	run_policy("sc1", read_rows_sc1<8, 8>, 8, 8, in, out, bytes, 16);
	run_policy("sc0 sc1", read_rows_sc0_sc1<8, 8>, 8, 8, in, out, bytes, 16);
	run_policy("sc0 nt", read_rows_sc0_nt<8, 8>, 8, 8, in, out, bytes, 16);
	run_policy("sc1 nt", read_rows_sc1_nt<4, 8>, 4, 8, in, out, bytes, 4);
	run_policy("sc0 sc1 nt", read_rows_sc0_sc1_nt<4, 8>, 4, 8, in, out, bytes, 4);
	return 0;
}
