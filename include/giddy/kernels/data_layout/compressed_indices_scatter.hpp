// Scatter ("patching") with COMPRESSED target positions, for CDNA4.
//
// Patch positions come as runs; run rho has a baseline, an offset width W in BITS (8/16/32/64), a
// length and the start of its offsets in a byte pool (in units of OutputIndexSize bytes):
//   target[baseline[rho] + offsets_rho[t]] = data[cursor(rho) + t],   t < run_lengths[rho]
// A new run starts at every multiple of `anchoring_period` patches; scatter_position_anchors[a] is
// the first run of patch block a, so cursor(rho) = a*anchoring_period + (lengths of the block's
// earlier runs).  Runs with any other W are skipped (their patches still advance the cursor).
// Same kernel template name, parameter list and semantics as the reference's
// src/kernels/data_layout/compressed_indices_scatter.cuh:219-271 (run loop :71-125, single run :31-54).
//
// The reference walks a block's runs one after another, each with a block-stride loop — a short
// run keeps one lane in 32+ busy.  Here a WAVEFRONT owns a patch block (blocks at grid stride):
// it loads 64 run descriptors at once (one per lane), prefix-sums their lengths with DPP to get
// every run's data cursor, then flattens the batch: lane-strided over ALL the batch's patches,
// each patch finding its run by a 6-step binary search over the 64 cursors held in LDS.  Work is
// balanced whatever the run lengths.  Patches are ~0.1 % of a column: the kernel is tiny next to
// the decoder it follows on the stream.
#pragma once
#include <hip/hip_runtime.h>

#include <stdexcept>

#include "giddy/common.hpp"
#include "giddy/kernels/launch_config_resolution_params.hpp"
#include "giddy/on_device/wave64.hpp"
#include "giddy/util/math.hpp"

namespace cuda {
namespace kernels {
namespace scatter {
namespace compressed_indices {

namespace detail {

enum : unsigned { max_waves_per_block = 16 };

struct run_descriptor {
	uint64_t cursor;     // position of the run's first patch, relative to the patch block's first patch
	uint64_t baseline;
	uint64_t offsets;    // byte offset of the run's offsets within the pool
	uint32_t length;     // clamped to the patches left in the block (a run cannot be longer)
	uint32_t width;      // bytes per offset, 0 = skip
};

/// The walk over the compressed positions, shared by the scatter and by the stand-alone expansion of the positions:
/// calls sink(patch index, position, representable) once per patch, consecutive lanes taking consecutive patches.
template <unsigned OutputIndexSize, unsigned InputIndexSize, unsigned RunLengthSize, typename Sink>
__device__ __forceinline__ void for_each_patch(
	const uint_t<RunLengthSize>*    __restrict__  scatter_position_run_lengths,
	const unsigned char*            __restrict__  scatter_position_run_individual_offset_sizes,
	const uint_t<OutputIndexSize>*  __restrict__  scatter_position_run_baseline_values,
	const uint_t<InputIndexSize>*   __restrict__  scatter_position_run_offsets_start_positions,
	const unsigned char*            __restrict__  scatter_position_offset_bytes,
	const uint_t<InputIndexSize>*   __restrict__  scatter_position_anchors,
	uint64_t period, uint64_t num_runs, uint64_t n, run_descriptor* mine, Sink sink)
{
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) blockIdx.x * wave64::per_block() + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) gridDim.x * wave64::per_block();
	if (period == 0) { return; }
	const uint64_t num_blocks = (n + period - 1) / period;

	for (uint64_t a = wave; a < num_blocks; a += num_waves) {
		const uint64_t block_first_patch = a * period;
		const uint64_t block_patches = (block_first_patch + period < n) ? period : n - block_first_patch;
		const uint64_t first_run = scatter_position_anchors[a];
		const uint64_t end_run = ((a + 1) * period >= n) ? num_runs : (uint64_t) scatter_position_anchors[a + 1];
		uint64_t consumed = 0;   // patches of this block covered by earlier batches

		for (uint64_t batch = first_run; batch < end_run && consumed < block_patches; batch += 64) {
			const uint64_t rho = batch + lane;
			const bool have = rho < end_run;
			uint64_t length = have ? (uint64_t) scatter_position_run_lengths[rho] : 0;
			if (length > block_patches) { length = block_patches; }
			const uint32_t clamped = (uint32_t) length;   // block_patches <= period, assumed < 2^32 per batch of 64
			uint64_t inclusive = wave64::inclusive_sum((uint64_t) clamped);
			run_descriptor d;
			d.cursor = consumed + inclusive - clamped;
			d.length = clamped;
			d.baseline = have ? (uint64_t) scatter_position_run_baseline_values[rho] : 0;
			d.offsets = have ? (uint64_t) scatter_position_run_offsets_start_positions[rho] * OutputIndexSize : 0;
			unsigned bits = have ? scatter_position_run_individual_offset_sizes[rho] : 0;
			d.width = (bits == 8 || bits == 16 || bits == 32 || bits == 64) ? bits / 8 : 0;
			mine[lane] = d;
			const uint64_t batch_total = wave64::read_last_lane(inclusive);
			wave64::lds_fence();

			uint64_t batch_end = consumed + batch_total;
			if (batch_end > block_patches) { batch_end = block_patches; }
			for (uint64_t k = consumed + lane; k < batch_end; k += 64) {
				// the run containing patch k: the last descriptor with cursor <= k and length > 0 ... cursors are
				// non-decreasing, so binary search for the last cursor <= k, then step over empty runs backwards
				unsigned lo = 0;
				#pragma unroll
				for (unsigned step = 32; step > 0; step >>= 1) { if (mine[lo + step].cursor <= k) { lo += step; } }
				while (lo > 0 && mine[lo].length == 0) { lo--; }
				const run_descriptor r = mine[lo];
				const uint64_t t = k - r.cursor;
				if (t < r.length) {
					uint64_t offset = 0;
					if (r.width != 0) {
						const unsigned char* where = scatter_position_offset_bytes + r.offsets + t * r.width;
						switch (r.width) {
						case 1: offset = *where; break;
						case 2: offset = *reinterpret_cast<const uint16_t*>(where); break;
						case 4: offset = *reinterpret_cast<const uint32_t*>(where); break;
						default: offset = *reinterpret_cast<const uint64_t*>(where); break;
						}
					}
					sink(block_first_patch + k, (uint_t<OutputIndexSize>) (r.baseline + offset), r.width != 0);
				}
			}
			consumed += batch_total;
			wave64::lds_fence();
		}
	}
}

} // namespace detail

template <unsigned OutputIndexSize, unsigned ElementSize, unsigned InputIndexSize, unsigned RunLengthSize = InputIndexSize>
__global__ void scatter(
	uint_t<ElementSize>*            __restrict__  target,
	const uint_t<ElementSize>*      __restrict__  data,
	const uint_t<RunLengthSize>*    __restrict__  scatter_position_run_lengths,
	const unsigned char*            __restrict__  scatter_position_run_individual_offset_sizes,
	const uint_t<OutputIndexSize>*  __restrict__  scatter_position_run_baseline_values,
	const uint_t<InputIndexSize>*   __restrict__  scatter_position_run_offsets_start_positions,
	const unsigned char*            __restrict__  scatter_position_offset_bytes,
	const uint_t<InputIndexSize>*   __restrict__  scatter_position_anchors,
	uint_t<InputIndexSize>                        anchoring_period,
	uint_t<InputIndexSize>                        num_output_position_runs,
	size_type_by_index_size<InputIndexSize>       data_length)
{
	__shared__ detail::run_descriptor descriptors[detail::max_waves_per_block][64];
	detail::for_each_patch<OutputIndexSize, InputIndexSize, RunLengthSize>(
		scatter_position_run_lengths, scatter_position_run_individual_offset_sizes, scatter_position_run_baseline_values,
		scatter_position_run_offsets_start_positions, scatter_position_offset_bytes, scatter_position_anchors,
		anchoring_period, num_output_position_runs, data_length, descriptors[wave64::index_in_block() % detail::max_waves_per_block],
		[&](uint64_t patch, uint_t<OutputIndexSize> position, bool representable) {
			if (representable) { target[position] = data[patch]; }
		});
}

/// The positions alone, as a plain array (positions[k] = where patch k goes): what the fused decoders merge from
/// (kernel_wrappers/decompression/patched/*.hpp) — bytes proportional to the patches, not to the column.
/// A run whose offset width is none of 8/16/32/64 bits — which scatter() skips, compressed_indices_scatter.cuh:105-122 —
/// has NO plain-array equivalent: its patches are written as all-ones, which a fused decoder reads as "no more patches"
/// (later patches of that wavefront's chunk are then not applied).  Such input is outside the fused wrappers' contract;
/// `unsupported_patches` (may be null) counts the patches of such runs so that a caller can find out.
template <unsigned OutputIndexSize, unsigned InputIndexSize, unsigned RunLengthSize = InputIndexSize>
__global__ void expand_positions(
	uint_t<OutputIndexSize>*        __restrict__  positions,
	const uint_t<RunLengthSize>*    __restrict__  scatter_position_run_lengths,
	const unsigned char*            __restrict__  scatter_position_run_individual_offset_sizes,
	const uint_t<OutputIndexSize>*  __restrict__  scatter_position_run_baseline_values,
	const uint_t<InputIndexSize>*   __restrict__  scatter_position_run_offsets_start_positions,
	const unsigned char*            __restrict__  scatter_position_offset_bytes,
	const uint_t<InputIndexSize>*   __restrict__  scatter_position_anchors,
	uint_t<InputIndexSize>                        anchoring_period,
	uint_t<InputIndexSize>                        num_output_position_runs,
	size_type_by_index_size<InputIndexSize>       data_length,
	native_word_t*                  __restrict__  unsupported_patches = nullptr)
{
	__shared__ detail::run_descriptor descriptors[detail::max_waves_per_block][64];
	detail::for_each_patch<OutputIndexSize, InputIndexSize, RunLengthSize>(
		scatter_position_run_lengths, scatter_position_run_individual_offset_sizes, scatter_position_run_baseline_values,
		scatter_position_run_offsets_start_positions, scatter_position_offset_bytes, scatter_position_anchors,
		anchoring_period, num_output_position_runs, data_length, descriptors[wave64::index_in_block() % detail::max_waves_per_block],
		[&](uint64_t patch, uint_t<OutputIndexSize> position, bool representable) {
			positions[patch] = representable ? position : (uint_t<OutputIndexSize>) ~(uint_t<OutputIndexSize>) 0;
			if (!representable && unsupported_patches != nullptr) { atomicAdd(unsupported_patches, 1u); }
		});
}

template <unsigned OutputIndexSize, unsigned ElementSize, unsigned InputIndexSize, unsigned RunLengthSize = InputIndexSize>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;

	launch_config_resolution_params_t(
		device::properties_t device_properties_, device_function::attributes_t kernel_function_attributes_,
		size_t input_data_length, size_t anchoring_period)
	: parent(device_properties_, kernel_function_attributes_, nullopt)
	{
		if (input_data_length == 0) { throw std::invalid_argument("Zero-length scatters not currently supported"); }
		if (anchoring_period == 0) { throw std::invalid_argument("The anchoring period must be a positive number"); }
		grid_construction_resolution  = warp;             // one wavefront per anchored block of patches
		length                        = util::div_rounding_up(input_data_length, anchoring_period);
		serialization_option          = auto_maximized;
		quanta.threads_in_block       = warp_size;
		upper_limits.threads_in_block = 256;
		keep_gpu_busy_factor          = 1024;  // in practice: one tile per workgroup — the dispatcher hands tiles out in address order, which HBM likes best
	}

	launch_config_resolution_params_t(device::properties_t device_properties_, size_t input_data_length, size_t anchoring_period)
	: launch_config_resolution_params_t(device_properties_,
		device_function_t(scatter<OutputIndexSize, ElementSize, InputIndexSize, RunLengthSize>).attributes(),
		input_data_length, anchoring_period) { }
};

} // namespace compressed_indices
} // namespace scatter
} // namespace kernels
} // namespace cuda
