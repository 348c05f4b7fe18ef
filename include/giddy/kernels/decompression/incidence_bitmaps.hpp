// Incidence-bitmaps decompression for CDNA4.
//
// One bitmap per distinct value; bit j of word w of bitmap b says "element 32*w + j has value b":
//   out[32*w + j] = dictionary_entries[b]   (or Uncompressed(b) when !UseDictionary)
// Bitmaps are concatenated at a stride of round_up(bitmap_length, BitmapAlignmentInInts) words;
// bitmap_length counts 32-bit words; the output has 32*bitmap_length elements; when several bitmaps
// claim an element the highest index wins.  Same kernel template name, parameter list and element
// semantics as the reference's src/kernels/decompression/incidence_bitmaps.cuh:69-131 (bit peel :33-61).
//
// The reference ties the data layout to a 32-lane warp (static_assert warp_size == 32 == bits per
// container, :83-84): each thread peels one container bit by bit into a per-thread 32-byte shared
// memory strip, then the warp copies the strip out through the dictionary.  Here, per wavefront and
// per 64 consecutive containers (2048 outputs):
//   1. lane l owns container l: it loads its word of EVERY bitmap (coalesced 256-byte wave loads, 16 in
//      flight) and folds them, in ascending bitmap order, into 8 bit-planes of the winning bitmap index
//      with one v_bfi_b32 per plane per bitmap ("highest index wins" falls out of the order);
//   2. the planes go through wave-private LDS (2 KiB) so that outputs can be written in order:
//      a lane owning E = 16/UncompressedSize consecutive outputs reads the 8 planes of their container
//      (two 16-byte LDS reads), turns each 4-bit slice of the planes into four byte indices with a
//      multiply-and-mask bit spread (3 VALU ops per plane), translates them through an LDS copy of the
//      (<= 256 entry) dictionary and issues 16-byte stores, contiguous across the wavefront.
// No __syncthreads after the dictionary is staged.  Elements covered by no bitmap are undefined in the
// reference (stale shared memory); here they decode as bitmap 0.
//
// HBM bytes per element: UncompressedSize + num_bitmaps/8.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "giddy/common.hpp"
#include "giddy/kernels/launch_config_resolution_params.hpp"
#include "giddy/on_device/wave64.hpp"
#include "giddy/util/math.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace incidence_bitmaps {

using bitmap_index_t   = uint8_t;
using bitmap_size_type = size_type_by_index_size<sizeof(bitmap_index_t)>;

namespace detail {

enum : unsigned { max_bitmaps = 256, bitmaps_in_flight = 16, planes = 8, containers_per_wave = 64,
	max_waves_per_block = 4 };   // (the launch parameters cap the workgroup at 256 threads; 16 wavefronts' planes — 32 KiB — held a CU to four workgroups)

struct alignas(16) plane_set { uint32_t plane[planes]; };   // bit j of plane[k] = bit k of the bitmap index of the container's element j

} // namespace detail


template <unsigned IndexSize, unsigned UncompressedSize, unsigned BitmapAlignmentInInts, bool UseDictionary = true>
__global__ void decompress(
	uint_t<UncompressedSize>*            __restrict__  decompressed,
	const standard_bit_container_t*      __restrict__  bitmaps,
	const uint_t<UncompressedSize>*      __restrict__  dictionary_entries,
	size_type_by_index_size<IndexSize>                 bitmap_length,
	bitmap_size_type                                   num_bitmaps)
{
	using namespace memory_access;
	using U = uint_t<UncompressedSize>;
	constexpr unsigned E = 16 / UncompressedSize;
	constexpr unsigned B = detail::bitmaps_in_flight;
	constexpr unsigned W = detail::containers_per_wave;
	static_assert(32 % E == 0, "a lane's elements must share a container");

	__shared__ U dictionary[detail::max_bitmaps];
	__shared__ detail::plane_set plane_sets[detail::max_waves_per_block][W];
	if (UseDictionary) {
		for (unsigned i = threadIdx.x; i < num_bitmaps && i < detail::max_bitmaps; i += blockDim.x) { dictionary[i] = dictionary_entries[i]; }
		__syncthreads();
	}

	// LDS is provisioned for max_waves_per_block wavefronts; in a larger workgroup the extra wavefronts sit idle (no barrier follows)
	const unsigned working_waves = wave64::per_block() < detail::max_waves_per_block ? wave64::per_block() : detail::max_waves_per_block;
	if (wave64::index_in_block() >= working_waves) { return; }
	const unsigned lane = wave64::lane_id();
	detail::plane_set* mine = plane_sets[wave64::index_in_block()];
	const uint64_t wave = (uint64_t) blockIdx.x * working_waves + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) gridDim.x * working_waves;
	const uint64_t words = bitmap_length;
	const uint64_t aligned_words = (words + BitmapAlignmentInInts - 1) / BitmapAlignmentInInts * BitmapAlignmentInInts;
	const uint64_t num_chunks = (words + W - 1) / W;
	const bool aligned = is_aligned(decompressed, 16);

	// The index of the winning bitmap has as many bits as the bitmaps need: a column of 16 distinct values folds and unfolds 4 bit-planes, not 8
	// (round 4: the 1-byte instantiation spent 752 instructions per 2 KiB of output, most of them on planes that were zero: 68 % of the HBM peak).
	auto with_planes = [&](auto planes_in_use) {
		constexpr unsigned P = decltype(planes_in_use)::value;
		for (uint64_t chunk = wave; chunk < num_chunks; chunk += num_waves) {
			// 1. fold this lane's container of every bitmap into index bit-planes
			const uint64_t w = chunk * W + lane;
			const bool have = w < words;
			uint32_t folded[P];
			#pragma unroll
			for (unsigned k = 0; k < P; k++) { folded[k] = 0; }
			for (unsigned b0 = 0; b0 < num_bitmaps; b0 += B) {
				standard_bit_container_t word[B];
				#pragma unroll
				for (unsigned i = 0; i < B; i++) { word[i] = (have && b0 + i < num_bitmaps) ? bitmaps[(uint64_t) (b0 + i) * aligned_words + w] : 0u; }
				#pragma unroll
				for (unsigned i = 0; i < B; i++) {
					const unsigned b = b0 + i;   // wave-uniform
					#pragma unroll
					for (unsigned k = 0; k < P; k++) {
						const uint32_t select = ((b >> k) & 1u) ? 0xFFFFFFFFu : 0u;
						folded[k] = (word[i] & select) | (~word[i] & folded[k]);   // v_bfi_b32
					}
				}
			}
			#pragma unroll
			for (unsigned k = 0; k < P; k++) { mine[lane].plane[k] = folded[k]; }
			wave64::lds_fence();

			// 2. outputs in order: E consecutive elements per lane per step
			const uint64_t chunk_first_element = chunk * W * 32;
			const uint64_t chunk_elements = (words - chunk * W < W ? words - chunk * W : W) * 32;
			#pragma unroll 2
			for (unsigned step = 0; step < (W * 32) / (64 * E); step++) {
				const unsigned e0 = step * 64 * E + lane * E;
				if (e0 < chunk_elements) {
					uint32_t p[P];
					#pragma unroll
					for (unsigned k = 0; k < P; k++) { p[k] = mine[e0 >> 5].plane[k]; }
					const unsigned shift = e0 & 31;
					group<U, E> values;
					#pragma unroll
					for (unsigned c = 0; c < (E + 3) / 4; c++) {
						uint32_t indices = 0;   // byte j = bitmap index of element 4c + j
						#pragma unroll
						for (unsigned k = 0; k < P; k++) {
							const uint32_t nibble = (p[k] >> (shift + 4 * c)) & 0xFu;
							indices |= ((nibble * 0x00204081u) & 0x01010101u) << k;   // bit j of the nibble -> bit k of byte j
						}
						#pragma unroll
						for (unsigned j = 0; j < 4 && 4 * c + j < E; j++) {
							const unsigned index = (indices >> (8 * j)) & 0xFFu;
							values[4 * c + j] = UseDictionary ? dictionary[index] : (U) index;
						}
					}
					U* where = decompressed + chunk_first_element + e0;
					if (aligned) { store_group<memory_access::wide_output_policy<UncompressedSize>(memory_access::streaming)>(where, values); }   // (write-through: +3 / +7 points for 4- / 8-byte elements)
					else {
						#pragma unroll
						for (unsigned j = 0; j < E; j++) { where[j] = values[j]; }
					}
				}
			}
			wave64::lds_fence();   // the planes are overwritten by the next chunk
		}
	};
	if (num_bitmaps <= 4) { with_planes(std::integral_constant<unsigned, 2>()); }
	else if (num_bitmaps <= 16) { with_planes(std::integral_constant<unsigned, 4>()); }
	else if (num_bitmaps <= 64) { with_planes(std::integral_constant<unsigned, 6>()); }
	else { with_planes(std::integral_constant<unsigned, 8>()); }
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned BitmapAlignmentInInts, bool UseDictionary = true>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;

	/// `uncompressed_length` counts ELEMENTS (as the reference's wrapper passes it, incidence_bitmaps.cu:54)
	launch_config_resolution_params_t(
		device::properties_t device_properties_, device_function::attributes_t kernel_function_attributes_,
		size_t uncompressed_length, optional<shared_memory_size_t> dynamic_shared_mem_limit = nullopt)
	: parent(device_properties_, kernel_function_attributes_, dynamic_shared_mem_limit)
	{
		grid_construction_resolution  = warp;   // one wavefront per 64 containers (2048 elements)
		length                        = util::div_rounding_up(uncompressed_length, (size_t) (32 * detail::containers_per_wave));
		serialization_option          = auto_maximized;
		quanta.threads_in_block       = warp_size;
		upper_limits.threads_in_block = 256;
		keep_gpu_busy_factor          = 1024;
	}

	launch_config_resolution_params_t(
		device::properties_t device_properties_, size_t uncompressed_length,
		optional<shared_memory_size_t> dynamic_shared_mem_limit = nullopt)
	: launch_config_resolution_params_t(device_properties_,
		device_function_t(decompress<IndexSize, UncompressedSize, BitmapAlignmentInInts, UseDictionary>).attributes(),
		uncompressed_length, dynamic_shared_mem_limit) { }
};

} // namespace incidence_bitmaps
} // namespace decompression
} // namespace kernels
} // namespace cuda
