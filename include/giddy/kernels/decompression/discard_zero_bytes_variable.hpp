// Discard-zero-bytes (null suppression), VARIABLE width, for CDNA4.
//
// Element i keeps only its n_i low-order bytes, stored back to back (little-endian) in
// compressed_data; n_i = min_represented_element_size + f_i, where f_i is a bits_per_element_size-bit
// field, fields packed LSB-first into containers of ElementSizesContainerSize bytes,
// floor(container bits / field bits) per container, never straddling (high slack bits unused);
// position_anchors[s] is the byte offset of element s*segment_length.
//
// Same kernel template name, parameter list and element semantics as the reference's
// src/kernels/decompression/discard_zero_bytes_variable.cuh:381-439 (get_element_size :44-58,
// pad_high_bytes :196-213, warp body :241-330) — except that every tail length is handled (the
// reference's last-warp test is only right when length % segment_length is 0 or <= 32, quirk a8).
//
// Execution.  The reference gives a 32-lane warp a whole segment and moves one element per lane per
// iteration: integer div+mod, a 5-step shuffle scan, a byte-wise unaligned memcpy from global
// memory and a broadcast, per 32 elements.  Here a WAVEFRONT owns a segment (segments at grid
// stride) and works in rows of 64*E elements, E = 16/UncompressedSize consecutive elements per lane:
//   1. size fields: one division per lane per row, then E shifts/masks;
//   2. byte offsets: in-lane running sum + one 64-lane DPP prefix sum (row byte counts < 2^11);
//   3. the row's compressed bytes — one contiguous range of <= 1 KiB — are staged into LDS with
//      16-byte-per-lane COALESCED loads of exactly the 16-byte chunks that overlap the range (so the
//      kernel never touches a chunk holding no valid byte: nothing is read past the buffer's end);
//   4. each element is cut out of LDS with aligned 32-bit reads + a funnel shift and masked to n_i bytes;
//   5. one 16-byte store per lane, contiguous across the wavefront.
// Rows are processed four at a time so that four rows' requests are in flight behind each dependent
// step (size words -> byte ranges -> elements).  No __syncthreads; wavefronts are independent.  HBM bytes per element: mean n_i + field bits/8 +
// UncompressedSize + anchor/segment_length  (10.76 B for 1..4-byte sizes into 64 bits, config C5).
#pragma once
#include <hip/hip_runtime.h>

#include "giddy/common.hpp"
#include "giddy/kernels/launch_config_resolution_params.hpp"
#include "giddy/on_device/wave64.hpp"
#include "giddy/util/math.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace discard_zero_bytes {
namespace variable_width {

using element_size_t = native_word_t;

namespace detail {

enum : unsigned {
	max_waves_per_block = 4,          // the launch parameters cap the workgroup at 256 threads
	rows_in_flight = 4,
	staging_bytes = 1024 + 16 + 16,   // a row's bytes + head misalignment + over-read slack for the aligned dword reads
	staging_words = staging_bytes / 4,
};

} // namespace detail

template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
__global__ void decompress(
	uint_t<UncompressedSize>*  __restrict__  decompressed,
	const uint8_t*             __restrict__  compressed_data,
	const uint_t<ElementSizesContainerSize>*
	                           __restrict__  packed_element_sizes,
	const uint_t<IndexSize>*   __restrict__  position_anchors,
	size_type_by_index_size<IndexSize>       segment_length,
	size_type_by_index_size<IndexSize>       length,
	element_size_t                           min_represented_element_size,
	native_word_t                            bits_per_element_size)
{
	using namespace memory_access;
	using U = uint_t<UncompressedSize>;
	using container_type = uint_t<ElementSizesContainerSize>;
	static_assert(ElementSizesContainerSize == 4 || ElementSizesContainerSize == 8,
		"Only container types with 32 or 64 bits are currently supported");
	constexpr unsigned E = 16 / UncompressedSize;
	constexpr unsigned row = 64 * E;
	using chunk_t = typename chunk<16>::type;

	constexpr unsigned R = detail::rows_in_flight;
	__shared__ __attribute__((aligned(16))) uint32_t staging[detail::max_waves_per_block][R * detail::staging_words];
	// staging exists for max_waves_per_block wavefronts; in a larger workgroup the extra wavefronts sit idle
	const unsigned working_waves = wave64::per_block() < detail::max_waves_per_block ? wave64::per_block() : detail::max_waves_per_block;
	if (wave64::index_in_block() >= working_waves) { return; }
	uint32_t* stage = staging[wave64::index_in_block()];

	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) blockIdx.x * working_waves + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) gridDim.x * working_waves;
	const uint64_t L = segment_length, n = length;
	if (L == 0 || bits_per_element_size == 0) { return; }
	const uint64_t num_segments = (n + L - 1) / L;
	const unsigned bits = bits_per_element_size;
	const unsigned per = (8 * ElementSizesContainerSize) / bits;
	const container_type field_mask = bits >= 8 * ElementSizesContainerSize ? ~(container_type) 0 : (((container_type) 1 << bits) - 1);

	for (uint64_t s = wave; s < num_segments; s += num_waves) {
		const uint64_t segment_start = s * L;
		const uint64_t segment_end = (segment_start + L < n) ? segment_start + L : n;
		uint64_t bytes_start = position_anchors[s];   // byte offset of the next row's first element

		// R rows per iteration: all size words first, then all byte ranges, then all extractions, so that
		// R rows' worth of HBM requests are in flight behind each dependent step
		for (uint64_t rows_start = segment_start; rows_start < segment_end; rows_start += (uint64_t) R * row) {
			unsigned sizes[R][E]; unsigned offset[R]; unsigned row_bytes[R];
			// 1. element sizes of R rows
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				const uint64_t i0 = rows_start + (uint64_t) k * row + (uint64_t) lane * E;
				uint64_t container_index; unsigned field;
				if (IndexSize <= 4 || (i0 >> 32) == 0) { uint32_t q = (uint32_t) i0 / per; container_index = q; field = (uint32_t) i0 - q * per; }
				else { container_index = i0 / per; field = (unsigned) (i0 - container_index * per); }
				unsigned lane_bytes = 0;
				container_type container = (i0 < segment_end) ? packed_element_sizes[container_index] : (container_type) 0;
				#pragma unroll
				for (unsigned j = 0; j < E; j++) {
					unsigned size = 0;
					if (i0 + j < segment_end) {
						size = min_represented_element_size + (unsigned) ((container >> (field * bits)) & field_mask);
						if (size > UncompressedSize) { size = UncompressedSize; }   // precondition n_i <= sizeof(element); never read more
					}
					sizes[k][j] = size; lane_bytes += size;
					if (++field == per) { field = 0; container_index++; if (j + 1 < E && i0 + j + 1 < segment_end) { container = packed_element_sizes[container_index]; } }
				}
				offset[k] = lane_bytes;
			}
			// 2. byte offsets within each row (one DPP prefix sum per row; row byte counts < 2^11)
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				const unsigned inclusive = wave64::inclusive_sum(offset[k]);
				row_bytes[k] = wave64::read_last_lane(inclusive);
				offset[k] = inclusive - offset[k];
			}
			// 3. stage each row's bytes: the 16-byte chunks overlapping [address, address + row_bytes)
			uint64_t row_bytes_start = bytes_start;
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				const uintptr_t address = reinterpret_cast<uintptr_t>(compressed_data) + row_bytes_start;
				const uintptr_t first_chunk = address & ~(uintptr_t) 15;
				const unsigned head = (unsigned) (address - first_chunk);
				const unsigned num_chunks = row_bytes[k] == 0 ? 0 : (head + row_bytes[k] + 15) / 16;
				for (unsigned c = lane; c < num_chunks; c += 64) {
					reinterpret_cast<chunk_t*>(stage + k * detail::staging_words)[c] = *reinterpret_cast<const chunk_t*>(first_chunk + 16 * (uintptr_t) c);
				}
				offset[k] += head;
				row_bytes_start += row_bytes[k];
			}
			bytes_start = row_bytes_start;
			wave64::lds_fence();

			// 4. cut the elements out, 5. store
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				const uint64_t i0 = rows_start + (uint64_t) k * row + (uint64_t) lane * E;
				const uint32_t* row_stage = stage + k * detail::staging_words;
				group<U, E> values;
				unsigned at = offset[k];
				#pragma unroll
				for (unsigned j = 0; j < E; j++) {
					const unsigned word = at >> 2, shift = (at & 3) * 8, size = sizes[k][j];
					uint64_t v = 0;
					if (size != 0) {
						uint64_t low = (uint64_t) row_stage[word] | ((uint64_t) row_stage[word + 1] << 32);
						v = low >> shift;
						if (UncompressedSize == 8 && shift != 0) { v |= (uint64_t) row_stage[word + 2] << (64 - shift); }
						if (size < 8) { v &= (((uint64_t) 1) << (8 * size)) - 1; }
					}
					values[j] = (U) v;
					at += size;
				}
				if (i0 + E <= segment_end && is_aligned(decompressed + i0, 16)) { store_group(decompressed + i0, values); }
				else {
					#pragma unroll
					for (unsigned j = 0; j < E; j++) { if (i0 + j < segment_end) { decompressed[i0 + j] = values[j]; } }
				}
			}
			wave64::lds_fence();   // the staging area is rewritten by the next iteration
		}
	}
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;

	launch_config_resolution_params_t(
		device::properties_t device_properties_, device_function::attributes_t kernel_function_attributes_,
		size_t length_in_elements, size_t segment_length)
	: parent(device_properties_, kernel_function_attributes_)
	{
		if (segment_length == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
		grid_construction_resolution  = warp;              // one wavefront per anchored segment
		length                        = util::div_rounding_up(length_in_elements, segment_length);
		serialization_option          = auto_maximized;    // segments at grid stride
		quanta.threads_in_block       = warp_size;
		upper_limits.threads_in_block = 256;
		keep_gpu_busy_factor          = 1024;  // in practice: one tile per workgroup — the dispatcher hands tiles out in address order, which HBM likes best
	}

	launch_config_resolution_params_t(device::properties_t device_properties_, size_t length_in_elements, size_t segment_length)
	: launch_config_resolution_params_t(device_properties_,
		device_function_t(decompress<IndexSize, UncompressedSize, ElementSizesContainerSize>).attributes(),
		length_in_elements, segment_length) { }
};

} // namespace variable_width
} // namespace discard_zero_bytes
} // namespace decompression
} // namespace kernels
} // namespace cuda
