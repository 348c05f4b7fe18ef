// Model decompression:   out[p] = Uncompressed(model(p)),  one coefficient record passed BY VALUE.
//
// Kernel template name, parameter list and semantics as in the reference's
// src/kernels/decompression/model.cuh:38-53.  Nothing is read from memory: the kernel is a pure
// 16-byte-per-lane store stream (G = 16/sizeof(Uncompressed) consecutive elements per lane per
// step, K steps per tile, tiles at grid stride).  HBM bytes per element: sizeof(Uncompressed).
// decompress_with_intervals (model.cuh:55-98: wrapper-less in the reference) is below;
// it is launched with the same parameters class.
#pragma once
#include <hip/hip_runtime.h>

#include "giddy/common.hpp"
#include "giddy/functors/model_functors.hpp"
#include "giddy/kernels/launch_config_resolution_params.hpp"
#include "giddy/kernels/tuning.hpp"
#include "giddy/on_device/sorted_patches.hpp"
#include "giddy/on_device/stream_tiles.hpp"
#include "giddy/on_device/wave64.hpp"
#include "giddy/util/math.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace model {

template <unsigned IndexSize, typename Uncompressed, typename UnaryModelFunction>
__global__ void decompress(
	Uncompressed* __restrict__                      decompressed,
	typename UnaryModelFunction::coefficients_type  model_coefficients,
	size_type_by_index_size<IndexSize>              length)
{
	using model_index_type = typename UnaryModelFunction::index_type;
	constexpr unsigned G = stream_tiles::group_size<Uncompressed>::value;
	UnaryModelFunction model(model_coefficients);
	stream_tiles::generate<Uncompressed, false, GIDDY_MODEL_STEPS_PER_TILE, memory_access::wide_output_policy<sizeof(Uncompressed)>(GIDDY_MODEL_STORE_POLICY)>(decompressed, length,
		[&](uint64_t first_position, memory_access::group<Uncompressed, G>& out) {
			#pragma unroll
			for (unsigned j = 0; j < G; j++) { out[j] = static_cast<Uncompressed>(model((model_index_type) (first_position + j))); }
		},
		[&](uint64_t position) { return static_cast<Uncompressed>(model((model_index_type) position)); });
}

namespace patched {

/// Model decompression and patching in one pass: what `model::decompress` followed by
/// `data_layout::scatter::scatter(decompressed, patch_values, patch_positions, num_patches)` leaves in `decompressed`,
/// for ASCENDING patch positions (equal positions: the last one wins) — the usual way a modelled column is stored:
/// the model plus the elements it misses.
template <unsigned IndexSize, typename Uncompressed, typename UnaryModelFunction, bool ManyPatches = false>
__global__ void decompress(
	Uncompressed* __restrict__                      decompressed,
	typename UnaryModelFunction::coefficients_type  model_coefficients,
	const Uncompressed* __restrict__                patch_values,
	const uint_t<IndexSize>* __restrict__           patch_positions,
	size_type_by_index_size<IndexSize>              length,
	size_type_by_index_size<IndexSize>              num_patches)
{
	using model_index_type = typename UnaryModelFunction::index_type;
	constexpr unsigned G = stream_tiles::group_size<Uncompressed>::value;
	__shared__ sorted_patches::window_storage<IndexSize, Uncompressed> patch_windows;
	__shared__ sorted_patches::row_storage<Uncompressed, ManyPatches ? 4 : 0, 16 / sizeof(Uncompressed), IndexSize> patch_rows;
	UnaryModelFunction model(model_coefficients);
	stream_tiles::generate_and_patch<Uncompressed, GIDDY_MODEL_PATCHED_ROWS_PER_CHUNK>(decompressed, length,
		[&](uint64_t first_position, memory_access::group<Uncompressed, G>& out) {
			#pragma unroll
			for (unsigned j = 0; j < G; j++) { out[j] = static_cast<Uncompressed>(model((model_index_type) (first_position + j))); }
		},
		[&](uint64_t position) { return static_cast<Uncompressed>(model((model_index_type) position)); },
		sorted_patches::cursor_t<IndexSize, Uncompressed, ManyPatches>(patch_values, patch_positions, num_patches, length, patch_windows, patch_rows));
}

} // namespace patched

/// Piecewise models:  out[p] = model_{p / modeling_period}(p % modeling_period)  — one coefficient record per
/// interval of `modeling_period` elements, the model evaluated at the position WITHIN the interval.  Kernel
/// template name and parameter list as the reference's wrapper-less decompress_with_intervals
/// (src/kernels/decompression/model.cuh:55-98): this is Frame of Reference without the offsets —
/// the missing piece for segment-wise Model + patches columns.  `intervals_per_block` is accepted for signature
/// parity; the tiling here does not depend on it.  (The reference's one-interval-per-block branch writes a whole
/// period even past `length`, model.cuh:84-92; here nothing is written at or after `length`.)
template <unsigned IndexSize, typename Uncompressed, typename UnaryModelFunction>
__global__ void decompress_with_intervals(
	Uncompressed* __restrict__          decompressed,
	const typename UnaryModelFunction::coefficients_type* __restrict__
	                                    interval_model_coefficients,
	size_type_by_index_size<IndexSize>  modeling_period,
	size_type_by_index_size<IndexSize>  intervals_per_block,
	size_type_by_index_size<IndexSize>  length)
{
	using model_index_type = typename UnaryModelFunction::index_type;
	using coefficients_type = typename UnaryModelFunction::coefficients_type;
	constexpr unsigned G = stream_tiles::group_size<Uncompressed>::value;
	(void) intervals_per_block;
	const uint64_t period = modeling_period;
	if (period == 0) { return; }

	auto split = [period](uint64_t p, uint64_t& interval, uint64_t& within) {
		if (IndexSize <= 4 || ((p | period) >> 32) == 0) {
			uint32_t q = (uint32_t) p / (uint32_t) period;
			interval = q; within = (uint32_t) p - q * (uint32_t) period;
		}
		else { interval = p / period; within = p - interval * period; }
	};
	stream_tiles::generate<Uncompressed, true, GIDDY_MODEL_STEPS_PER_TILE, memory_access::wide_output_policy<sizeof(Uncompressed)>(GIDDY_MODEL_STORE_POLICY)>(decompressed, length,
		[&](uint64_t first_position, memory_access::group<Uncompressed, G>& out) {
			uint64_t interval, within;
			split(first_position, interval, within);
			coefficients_type record = interval_model_coefficients[interval];   // once per group, again only across a boundary
			#pragma unroll
			for (unsigned j = 0; j < G; j++) {
				out[j] = static_cast<Uncompressed>(UnaryModelFunction(record)((model_index_type) within));
				if (++within == period) {
					within = 0; interval++;
					if (j + 1 < G) { record = interval_model_coefficients[interval]; }
				}
			}
		},
		[&](uint64_t position) {
			uint64_t interval, within;
			split(position, interval, within);
			return static_cast<Uncompressed>(UnaryModelFunction(interval_model_coefficients[interval])((model_index_type) within));
		});
}

template <unsigned IndexSize, typename Uncompressed, typename UnaryModelFunction>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;

	launch_config_resolution_params_t(
		device::properties_t device_properties_, device_function::attributes_t kernel_function_attributes_, size_t length_)
	: parent(device_properties_, kernel_function_attributes_)
	{
		constexpr unsigned elements_per_thread_per_tile = stream_tiles::elements_per_thread_per_tile<Uncompressed, GIDDY_MODEL_STEPS_PER_TILE>();
		grid_construction_resolution  = thread;
		length                        = util::div_rounding_up(length_, (size_t) elements_per_thread_per_tile);
		serialization_option          = auto_maximized;
		quanta.threads_in_block       = warp_size;
		upper_limits.threads_in_block = GIDDY_STREAM_THREADS_PER_BLOCK;
		keep_gpu_busy_factor          = 1024;  // in practice: one tile per workgroup — the dispatcher hands tiles out in address order, which HBM likes best
		default_serialization_factor  = warp_size / sizeof(Uncompressed);
		// a pure store stream likes shallow tiles (one 16-byte store per lane) and few wavefronts: 4 resident workgroups per CU
		// reach 83-89 % of HBM peak where 8 reach 77 and four-store tiles 71-78 (tools/occupancy_sweep.py)
		if (sizeof(Uncompressed) <= 4) { resident_blocks_per_multiprocessor_cap = 4u; }
	}

	launch_config_resolution_params_t(device::properties_t device_properties_, size_t length_)
	: launch_config_resolution_params_t(device_properties_,
		device_function_t(decompress<IndexSize, Uncompressed, UnaryModelFunction>).attributes(), length_) { }
};

} // namespace model
} // namespace decompression
} // namespace kernels
} // namespace cuda
