// What this box sustains on the library's OWN streaming loop (giddy/on_device/stream_tiles.hpp: 16 bytes per lane, K steps a tile, non-temporal
// stores, one tile a workgroup) with the identity in place of a decoder — the measured ceiling bench.py reports beside the 8 TB/s paper peak
// ("a measured on-box copy ceiling").  Two shapes:
//   stream_copy      4 bytes in : 4 bytes out per element   (a plain copy)
//   stream_widen     1 byte  in : 4 bytes out per element   (Dictionary<4,4,1>'s and DZB-fixed<4,4,1>'s traffic without their arithmetic)
//   stream_widen_2_to_4 / _2_to_8   2 bytes in : 4 / 8 bytes out   (Frame of reference <4,int,short>'s; Delta <4,8,2>'s and DZB-fixed <4,8,2>'s)
// Measurement infrastructure behind ctypes (tools/stream_ceiling.py); not part of libgiddy_amd.so.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "giddy/on_device/stream_tiles.hpp"

namespace {

template <typename In, typename Out, int Stores>
__global__ void __launch_bounds__(256) identity_stream(Out* __restrict__ out, const In* __restrict__ in, uint64_t n)
{
	using namespace cuda;
	constexpr unsigned G = stream_tiles::group_size<Out>::value;
	stream_tiles::transform<In, Out, stream_tiles::steps_per_tile, Stores>(out, in, n,
		[](uint64_t, const memory_access::group<In, G>& a, memory_access::group<Out, G>& b) {
			#pragma unroll
			for (unsigned j = 0; j < G; j++) { b[j] = (Out) a[j]; }
		},
		[](uint64_t, In x) { return (Out) x; });
}

/// `dynamic_lds`: bytes of LDS a workgroup reserves and does not use — the decoders' residency throttle (resolve_launch_configuration.cpp: 26 KiB = 6 workgroups
/// per CU, the headline's launch); `stores`: memory_access::store_policy (1 = nt, 2 = sc1 nt: the decoders' of 4-byte and wider outputs)
template <typename In, typename Out>
int launch(void* out, const void* in, uint64_t n, void* stream, uint32_t dynamic_lds, int stores)
{
	constexpr uint64_t per_block = 256 * cuda::stream_tiles::elements_per_thread_per_tile<Out>();
	const uint64_t blocks = (n + per_block - 1) / per_block;
	if (blocks == 0) { return 0; }
	if (blocks > 0x7fffffffull) { return (int) hipErrorInvalidConfiguration; }
	auto kernel = stores == 2 ? identity_stream<In, Out, cuda::memory_access::streaming> : identity_stream<In, Out, cuda::memory_access::non_temporal>;
	if (dynamic_lds > 65536) { (void) hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) dynamic_lds); }
	kernel<<<dim3((unsigned) blocks), dim3(256), dynamic_lds, static_cast<hipStream_t>(stream)>>>(static_cast<Out*>(out), static_cast<const In*>(in), n);
	return (int) hipGetLastError();
}

} // namespace

extern "C" int stream_copy(void* out, const void* in, uint64_t elements_of_4_bytes, void* stream, uint32_t dynamic_lds, int stores) { return launch<uint32_t, uint32_t>(out, in, elements_of_4_bytes, stream, dynamic_lds, stores); }
extern "C" int stream_widen(void* out, const void* in, uint64_t elements, void* stream, uint32_t dynamic_lds, int stores) { return launch<uint8_t, uint32_t>(out, in, elements, stream, dynamic_lds, stores); }
extern "C" int stream_widen_2_to_4(void* out, const void* in, uint64_t elements, void* stream, uint32_t dynamic_lds, int stores) { return launch<uint16_t, uint32_t>(out, in, elements, stream, dynamic_lds, stores); }
extern "C" int stream_widen_2_to_8(void* out, const void* in, uint64_t elements, void* stream, uint32_t dynamic_lds, int stores) { return launch<uint16_t, uint64_t>(out, in, elements, stream, dynamic_lds, stores); }
