// Frame-of-reference decompression for CDNA4.
//
//   out[p] = Uncompressed( Uncompressed(model_s(p - s*L)) + compressed_input[p] ),   s = p / L
//
// Same kernel template name, parameter list and element semantics as the reference's
// src/kernels/decompression/frame_of_reference.cuh:35-80 (so a direct `cuda::launch` of
// `decompress<...>` keeps working), with a different execution shape:
//
//  * the reference gives one thread one element and ties the grid to the segmentation
//    (segments-per-block, a fixed block length for a single segment — which makes a 1-segment,
//    >1024-element column un-launchable, frame_of_reference.cuh:111-113);
//  * here the kernel is a plain HBM stream: a lane owns G = 16/sizeof(Uncompressed) consecutive
//    elements per step, reads them as one 4/8-byte load, writes them as one 16-byte store, and
//    the tile loop is the one all streaming decoders share (on_device/stream_tiles.hpp).  Segments only
//    matter for picking the coefficient record: one integer division and one record fetch (through
//    L1/L2; neighbouring lanes hit the same record) per G elements.  Any grid/block shape is correct.
//
// HBM traffic per element: sizeof(Compressed) read + sizeof(Uncompressed) written (+ one record per
// segment) — the algorithmic minimum (6 B for int32 <- int16).
#pragma once
#include <hip/hip_runtime.h>

#include <stdexcept>

#include "giddy/common.hpp"
#include "giddy/functors/model_functors.hpp"
#include "giddy/kernels/launch_config_resolution_params.hpp"
#include "giddy/kernels/tuning.hpp"
#include "giddy/on_device/stream_tiles.hpp"
#include "giddy/on_device/batch_pieces.hpp"
#include "giddy/on_device/wave64.hpp"
#include "giddy/util/math.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace frame_of_reference {

enum : serialization_factor_t { DefaultSerializationFactor = 16 };

namespace detail {

/// 16-byte stores per lane per tile: 2 where 16-bit offsets become 32-bit values (measured: +2...+4 points against 4; 8-bit offsets lose 3...9 with 2)
template <typename Uncompressed, typename Compressed>
constexpr unsigned steps_per_tile() { return (sizeof(Uncompressed) == 4 && sizeof(Compressed) == 2 && GIDDY_FOR_STEPS_PER_TILE == 4) ? 2 : GIDDY_FOR_STEPS_PER_TILE; }


template <unsigned IndexSize>
__device__ __forceinline__ void divide(uint64_t p, uint64_t l, uint64_t& quotient, uint64_t& remainder)
{
	if (IndexSize <= 4 || ((p | l) >> 32) == 0) {
		uint32_t q = (uint32_t) p / (uint32_t) l;
		quotient = q; remainder = (uint32_t) p - q * (uint32_t) l;
	}
	else { quotient = p / l; remainder = p - quotient * l; }
}

} // namespace detail

namespace detail {

template <unsigned IndexSize, typename Uncompressed, typename Compressed, typename UnaryModelFunction, typename Patches, bool AnyAlignment = false, typename Grid = launch_grid::whole>
__device__ __forceinline__ void decompress(
	Uncompressed*       __restrict__    decompressed,
	const Compressed*   __restrict__    compressed_input,
	const typename UnaryModelFunction::coefficients_type*
	                    __restrict__    segment_model_coefficients,
	uint64_t                            length,
	uint64_t                            L,
	Patches                             patches,
	Grid                                grid = Grid())   // (launch_grid.hpp: the launch's own, or a batch piece's part of it)
{
	using model_index_type = typename UnaryModelFunction::index_type;
	using coefficients_type = typename UnaryModelFunction::coefficients_type;
	constexpr unsigned G = stream_tiles::group_size<Uncompressed>::value;

	auto predict = [](const coefficients_type& record, uint64_t within, Compressed offset) -> Uncompressed {
		UnaryModelFunction segment_model(record);
		Uncompressed predicted_value = static_cast<Uncompressed>(segment_model((model_index_type) within));
		return (Uncompressed) (predicted_value + offset);
	};
	// A group's context: its first element's segment record and position within the segment.  When segments are whole
	// groups (segment_length % G == 0 — always so for 4-byte outputs, whose segment lengths are multiples of 4) a group
	// never crosses into the next segment and that one record is all it needs.
	struct group_context { coefficients_type record; uint64_t segment, within; };
	// (`AnyAlignment` — stream_tiles.hpp —: groups start at the output's first 128-byte boundary, wherever in a segment that is)
	const bool groups_stay_in_segment = L % G == 0 && (!AnyAlignment || stream_tiles::elements_before_first_line(decompressed, length) % G == 0);
	auto prepare = [&](uint64_t first_position) {
		group_context context;
		divide<IndexSize>(first_position, L, context.segment, context.within);
		context.record = segment_model_coefficients[context.segment];
		return context;
	};
	auto produce_group = [&](uint64_t, const memory_access::group<Compressed, G>& in, memory_access::group<Uncompressed, G>& out, const group_context& context) {
		if (groups_stay_in_segment) {
			#pragma unroll
			for (unsigned j = 0; j < G; j++) { out[j] = predict(context.record, context.within + j, in[j]); }
			return;
		}
		uint64_t segment = context.segment, within = context.within;
		coefficients_type record = context.record;
		#pragma unroll
		for (unsigned j = 0; j < G; j++) {
			out[j] = predict(record, within, in[j]);
			if (++within == L) {
				within = 0; segment++;
				if (j + 1 < G) { record = segment_model_coefficients[segment]; }
			}
		}
	};
	// (the patched decoder's row loop has no context phase: it prepares inside)
	auto produce_group_unprepared = [&](uint64_t first_position, const memory_access::group<Compressed, G>& in, memory_access::group<Uncompressed, G>& out) {
		produce_group(first_position, in, out, prepare(first_position));
	};
	auto produce_one = [&](uint64_t position, Compressed offset) {
		uint64_t segment, within;
		divide<IndexSize>(position, L, segment, within);
		return predict(segment_model_coefficients[segment], within, offset);
	};
	if constexpr (Patches::enabled) {   // (the plain decoder in this wave-chunk shape, with 4, 8 or 16 rows per chunk: -1...-7 points)
		// (8 rows in flight — a 4-byte column's whole chunk at once — measured 3-9 points faster than 4 for this decoder; Dictionary is best at 4)
		constexpr unsigned stage_up_to = GIDDY_FOR_PATCHED_STAGE_INPUT_UP_TO;
		constexpr bool stages = stream_tiles::stages_chunk_input<Compressed, Uncompressed, stage_up_to>();
		__shared__ stream_tiles::chunk_input_storage<Compressed, Uncompressed, stages> chunk_staging;
		stream_tiles::transform_and_patch<Compressed, Uncompressed, Patches::many ? GIDDY_FOR_MANY_PATCHED_ROWS_IN_FLIGHT : 8, stage_up_to>(decompressed, compressed_input, length, produce_group_unprepared, produce_one, patches,
			stream_tiles::nothing_to_do(), &chunk_staging, grid);
	}
	else {
		// (write-through stores where a segment's record is its one reference value: +1...2 points; with a line per segment -1...-4)
		constexpr int stores = sizeof(coefficients_type) == sizeof(Uncompressed) ? memory_access::wide_output_policy<sizeof(Uncompressed)>(GIDDY_FOR_STORE_POLICY) : (int) memory_access::non_temporal;
		stream_tiles::transform<Compressed, Uncompressed, steps_per_tile<Uncompressed, Compressed>(), stores, AnyAlignment>(decompressed, compressed_input, length, produce_group, produce_one, stream_tiles::nothing_to_do(), prepare, grid);
	}
}

} // namespace detail

template <unsigned IndexSize, typename Uncompressed, typename Compressed, typename UnaryModelFunction>
__global__ void decompress(
	Uncompressed*       __restrict__    decompressed,
	const Compressed*   __restrict__    compressed_input,
	const typename UnaryModelFunction::coefficients_type*
	                    __restrict__    segment_model_coefficients,
	size_type_by_index_size<IndexSize>  length,
	size_type_by_index_size<IndexSize>  segment_length,
	uint_t<IndexSize>                   num_segments_to_decompress_per_block) // kept for signature parity; the tiling here does not depend on it
{
	(void) num_segments_to_decompress_per_block;
	detail::decompress<IndexSize, Uncompressed, Compressed, UnaryModelFunction>(
		decompressed, compressed_input, segment_model_coefficients, length, segment_length, sorted_patches::none());
}

/// The same for a column at an address aligned for its elements only, or inside a 128-byte line (a slice [a, b) decoded into `decompressed + a`; `length` and the
/// segments count from `decompressed`): the `AnyAlignment` build of stream_tiles::transform — the wrapper launches it by the addresses (Dictionary: 26 -> 89 % of the HBM peak).
template <unsigned IndexSize, typename Uncompressed, typename Compressed, typename UnaryModelFunction>
__global__ void decompress_at_any_alignment(
	Uncompressed*       __restrict__    decompressed,
	const Compressed*   __restrict__    compressed_input,
	const typename UnaryModelFunction::coefficients_type*
	                    __restrict__    segment_model_coefficients,
	size_type_by_index_size<IndexSize>  length,
	size_type_by_index_size<IndexSize>  segment_length,
	uint_t<IndexSize>                   num_segments_to_decompress_per_block)
{
	(void) num_segments_to_decompress_per_block;
	detail::decompress<IndexSize, Uncompressed, Compressed, UnaryModelFunction, sorted_patches::none, true>(
		decompressed, compressed_input, segment_model_coefficients, length, segment_length, sorted_patches::none());
}

/// A piece of a batch launch (giddy/batch.hpp; launched as batch::decode_pieces<piece_body<...>>): `decompress`'s parameters out of the piece's record,
/// its body on the piece's part of the grid.  `AnyAlignment`: the build for pieces at any element offset (taken for the whole batch when one piece asks).
template <unsigned IndexSize, typename Uncompressed, typename Compressed, typename UnaryModelFunction, bool AnyAlignment = false>
struct piece_body {
	enum : unsigned { max_threads_per_block = 1024 };
	static __device__ __forceinline__ void run(const batch::piece_view piece, const launch_grid::part grid)
	{
		detail::decompress<IndexSize, Uncompressed, Compressed, UnaryModelFunction, sorted_patches::none, AnyAlignment>(
			piece.get<Uncompressed*>(0), piece.get<const Compressed*>(1), piece.get<const typename UnaryModelFunction::coefficients_type*>(2),
			piece.get<size_type_by_index_size<IndexSize>>(3), piece.get<size_type_by_index_size<IndexSize>>(4), sorted_patches::none(), grid);
	}
};

namespace patched {

/// Frame-of-reference decompression and patching in one pass: what `frame_of_reference::decompress`
/// followed by `data_layout::scatter::scatter(decompressed, patch_values, patch_positions, num_patches)` leaves in
/// `decompressed`, for ASCENDING patch positions (equal positions: the last one wins).
template <unsigned IndexSize, typename Uncompressed, typename Compressed, typename UnaryModelFunction, bool ManyPatches = false>
__global__ void decompress(
	Uncompressed*       __restrict__    decompressed,
	const Compressed*   __restrict__    compressed_input,
	const typename UnaryModelFunction::coefficients_type*
	                    __restrict__    segment_model_coefficients,
	const Uncompressed* __restrict__    patch_values,
	const uint_t<IndexSize>*
	                    __restrict__    patch_positions,
	size_type_by_index_size<IndexSize>  length,
	size_type_by_index_size<IndexSize>  segment_length,
	size_type_by_index_size<IndexSize>  num_patches)
{
	__shared__ sorted_patches::window_storage<IndexSize, Uncompressed> patch_windows;
	__shared__ sorted_patches::row_storage<Uncompressed, ManyPatches ? 4 : 0, 16 / sizeof(Uncompressed), IndexSize> patch_rows;
	detail::decompress<IndexSize, Uncompressed, Compressed, UnaryModelFunction>(
		decompressed, compressed_input, segment_model_coefficients, length, segment_length,
		sorted_patches::cursor_t<IndexSize, Uncompressed, ManyPatches>(patch_values, patch_positions, num_patches, length, patch_windows, patch_rows));
}

} // namespace patched

template <unsigned IndexSize, typename Uncompressed, typename Compressed, typename UnaryModelFunction>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;

	launch_config_resolution_params_t(
		device::properties_t           device_properties_,
		device_function::attributes_t  kernel_function_attributes_,
		size_t                         data_length,
		size_t                         segment_length)
	: parent(device_properties_, kernel_function_attributes_)
	{
		// same argument validation as frame_of_reference.cuh:99-104
		if (segment_length == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
		if (segment_length % sizeof(native_word_t) != 0) {
			throw std::invalid_argument("The segment length must be aligned with the GPU's native word size");
		}
		constexpr unsigned elements_per_thread_per_tile = stream_tiles::elements_per_thread_per_tile<Uncompressed, detail::steps_per_tile<Uncompressed, Compressed>()>();
		grid_construction_resolution  = thread;
		length                        = util::div_rounding_up(data_length, elements_per_thread_per_tile);
		serialization_option          = auto_maximized;   // tiles at grid stride
		quanta.threads_in_block       = warp_size;
		upper_limits.threads_in_block = GIDDY_STREAM_THREADS_PER_BLOCK;
		keep_gpu_busy_factor          = 1024;  // in practice: one tile per workgroup — the dispatcher hands tiles out in address order, which HBM likes best
		// 6 resident workgroups per CU are 1-9 points faster than 8, for every registered width (tools/occupancy_sweep.py)
		resident_blocks_per_multiprocessor_cap = 6u;
	}

	launch_config_resolution_params_t(device::properties_t device_properties_, size_t data_length, size_t segment_length)
	: launch_config_resolution_params_t(device_properties_,
		device_function_t(decompress<IndexSize, Uncompressed, Compressed, UnaryModelFunction>).attributes(),
		data_length, segment_length) { }
};

} // namespace frame_of_reference
} // namespace decompression
} // namespace kernels
} // namespace cuda
