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
// memory strip, then the warp copies the strip out through the dictionary.  Here a lane owns
// E = 16/UncompressedSize consecutive OUTPUT elements (E divides 32, so they live in one container):
// it walks the bitmaps in ascending order — 4 independent loads in flight — keeping, per element,
// the index of the last bitmap whose bit was set, translates through an LDS copy of the (<= 256
// entry) dictionary and issues one 16-byte store, contiguous across the wavefront.  The 32/E lanes
// sharing a container read the same address, which the memory pipeline serves as one request.
// Elements covered by no bitmap are undefined in the reference (stale shared memory); here they
// decode as bitmap 0.
//
// HBM bytes per element: UncompressedSize + num_bitmaps/8.
#pragma once
#include <hip/hip_runtime.h>

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

namespace detail { enum : unsigned { max_bitmaps = 256, bitmaps_in_flight = 4 }; }

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
	static_assert(32 % E == 0, "a lane's elements must share a container");

	__shared__ U dictionary[detail::max_bitmaps];
	if (UseDictionary) {
		for (unsigned i = threadIdx.x; i < num_bitmaps && i < detail::max_bitmaps; i += blockDim.x) { dictionary[i] = dictionary_entries[i]; }
		__syncthreads();
	}

	const uint64_t words = bitmap_length;
	const uint64_t aligned_words = (words + BitmapAlignmentInInts - 1) / BitmapAlignmentInInts * BitmapAlignmentInInts;
	const uint64_t n = words * 32;
	const uint64_t num_groups = n / E;   // n is a multiple of 32, hence of E
	const bool aligned = is_aligned(decompressed, 16);
	const uint64_t global_thread = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t num_threads = (uint64_t) gridDim.x * blockDim.x;

	for (uint64_t g = global_thread; g < num_groups; g += num_threads) {
		const uint64_t p = g * E;
		const uint64_t w = p >> 5;
		const unsigned shift = (unsigned) (p & 31);
		unsigned index[E];
		#pragma unroll
		for (unsigned j = 0; j < E; j++) { index[j] = 0; }
		unsigned b = 0;
		for (; b + B <= num_bitmaps; b += B) {
			standard_bit_container_t word[B];
			#pragma unroll
			for (unsigned i = 0; i < B; i++) { word[i] = bitmaps[(uint64_t) (b + i) * aligned_words + w]; }
			#pragma unroll
			for (unsigned i = 0; i < B; i++) {
				unsigned mine = word[i] >> shift;
				#pragma unroll
				for (unsigned j = 0; j < E; j++) { if ((mine >> j) & 1u) { index[j] = b + i; } }
			}
		}
		for (; b < num_bitmaps; b++) {
			unsigned mine = bitmaps[(uint64_t) b * aligned_words + w] >> shift;
			#pragma unroll
			for (unsigned j = 0; j < E; j++) { if ((mine >> j) & 1u) { index[j] = b; } }
		}
		group<U, E> values;
		#pragma unroll
		for (unsigned j = 0; j < E; j++) { values[j] = UseDictionary ? dictionary[index[j]] : (U) index[j]; }
		if (aligned) { store_group(decompressed + p, values); }
		else {
			#pragma unroll
			for (unsigned j = 0; j < E; j++) { decompressed[p + j] = values[j]; }
		}
	}
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
		grid_construction_resolution  = thread;
		length                        = util::div_rounding_up(uncompressed_length, (size_t) (16 / UncompressedSize));
		serialization_option          = auto_maximized;
		quanta.threads_in_block       = warp_size;
		upper_limits.threads_in_block = 256;
		keep_gpu_busy_factor          = 1024;  // in practice: one tile per workgroup — the dispatcher hands tiles out in address order, which HBM likes best
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
