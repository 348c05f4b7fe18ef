// Does gfx950 serve BYTE-unaligned ds_read_b64 / ds_read_b32 (LDS) and global_load_dwordx2 correctly, and at what cost?
// (development probe for the DZB-variable extraction: hipcc --offload-arch=gfx950 -O3 tools/lds_unaligned_probe.hip -o /tmp/p && /tmp/p)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint64_t lds_read_b64(unsigned byte_address)
{
	uint64_t v;
	asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_address) : "memory");
	return v;
}
__device__ __forceinline__ uint32_t lds_read_b32(unsigned byte_address)
{
	uint32_t v;
	asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(byte_address) : "memory");
	return v;
}

__global__ void check(unsigned* mismatches64, unsigned* mismatches32, unsigned stride)
{
	__shared__ __attribute__((aligned(16))) uint8_t bytes[8192];
	for (unsigned i = threadIdx.x; i < 8192; i += blockDim.x) { bytes[i] = (uint8_t) (i * 7 + (i >> 8) * 13 + 1); }
	__syncthreads();
	const unsigned base = (unsigned) (uintptr_t) bytes;   // LDS byte address (low 32 bits of the generic pointer's offset)
	unsigned bad64 = 0, bad32 = 0;
	for (unsigned shift = 0; shift < 16; shift++) {
		const unsigned at = threadIdx.x * stride + shift;
		if (at + 8 > 8192) { continue; }
		uint64_t expect = 0;
		for (unsigned b = 0; b < 8; b++) { expect |= (uint64_t) bytes[at + b] << (8 * b); }
		if (lds_read_b64(base + at) != expect) { bad64++; }
		if (lds_read_b32(base + at) != (uint32_t) expect) { bad32++; }
	}
	atomicAdd(mismatches64, bad64); atomicAdd(mismatches32, bad32);
}

template <int Mode>
__global__ void timing(uint64_t* out, unsigned stride, unsigned iterations)
{
	__shared__ __attribute__((aligned(16))) uint32_t words[4096 + 8];
	for (unsigned i = threadIdx.x; i < 4096 + 8; i += blockDim.x) { words[i] = i * 2654435761u; }
	__syncthreads();
	const uint8_t* bytes = reinterpret_cast<const uint8_t*>(words);
	typedef uint64_t __attribute__((aligned(1))) unaligned_u64;
	uint64_t acc = 0;
	unsigned at = (threadIdx.x & 63) * stride + (Mode == 1 ? 0 : 3);
	for (unsigned i = 0; i < iterations; i++) {
		#pragma unroll
		for (unsigned u = 0; u < 8; u++) {
			const unsigned a = (at + u * 320) & 8191;
			if (Mode <= 1) { acc += *reinterpret_cast<const unaligned_u64*>(bytes + a); }   // ds_read_b64 at a byte address
			else {   // three aligned dwords + v_alignbyte, as the decoder does today
				const uint32_t* w = words + (a >> 2);
				uint32_t lo = __builtin_amdgcn_alignbyte(w[1], w[0], a & 3), hi = __builtin_amdgcn_alignbyte(w[2], w[1], a & 3);
				acc += ((uint64_t) hi << 32) | lo;
			}
		}
		at += Mode == 1 ? 8 * stride : 5;
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void check_global(const uint8_t* data, unsigned n, unsigned* mismatches)
{
	unsigned bad = 0;
	for (unsigned at = blockIdx.x * blockDim.x + threadIdx.x; at + 8 <= n; at += gridDim.x * blockDim.x) {
		uint64_t got;
		asm volatile("global_load_dwordx2 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(data + at) : "memory");
		uint64_t expect = 0;
		for (unsigned b = 0; b < 8; b++) { expect |= (uint64_t) data[at + b] << (8 * b); }
		if (got != expect) { bad++; }
	}
	atomicAdd(mismatches, bad);
}

int main()
{
	unsigned* counters; hipMalloc(&counters, 12); 
	for (unsigned stride : { 1u, 3u, 5u, 8u, 11u }) {
		hipMemset(counters, 0, 12);
		check<<<1, 256>>>(counters, counters + 1, stride);
		unsigned h[3]; hipMemcpy(h, counters, 12, hipMemcpyDeviceToHost);
		std::printf("LDS byte-unaligned reads, lane stride %2u bytes: ds_read_b64 mismatches %u, ds_read_b32 mismatches %u\n", stride, h[0], h[1]);
	}
	std::vector<uint8_t> host(1 << 20);
	for (size_t i = 0; i < host.size(); i++) { host[i] = (uint8_t) (i * 31 + (i >> 9) + 5); }
	uint8_t* data; hipMalloc(&data, host.size()); hipMemcpy(data, host.data(), host.size(), hipMemcpyHostToDevice);
	hipMemset(counters, 0, 12);
	check_global<<<64, 256>>>(data, (unsigned) host.size(), counters + 2);
	unsigned g; hipMemcpy(&g, counters + 2, 4, hipMemcpyDeviceToHost);
	std::printf("global_load_dwordx2 at every byte address of 1 MiB: mismatches %u\n", g);

	uint64_t* out; hipMalloc(&out, 1024 * 256 * 8);
	hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
	auto time_it = [&](auto kernel, const char* name, unsigned stride) {
		kernel<<<1024, 256>>>(out, stride, 2000); hipDeviceSynchronize();
		hipEventRecord(a); kernel<<<1024, 256>>>(out, stride, 2000); hipEventRecord(b); hipEventSynchronize(b);
		float ms; hipEventElapsedTime(&ms, a, b);
		std::printf("%-46s lane stride %2u: %.3f ms  (%.2f G wave-reads/s)\n", name, stride, ms, 1024.0 * 4 * 2000 * 8 / ms / 1e6);
	};
	for (unsigned stride : { 5u, 8u }) {
		time_it(timing<0>, "ds_read_b64, byte-unaligned", stride);
		time_it(timing<1>, "ds_read_b64, start aligned to the stride", stride);
		time_it(timing<2>, "3 aligned dwords + 2 v_alignbyte", stride);
	}
	return 0;
}
