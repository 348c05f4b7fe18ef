// Two-phase scan (prefix reduction) of UN-ANCHORED data, and its single-workgroup variant — the three
// kernels the reference keeps beside the decompressors without wrappers:
//
//   reduce_segments                 segment_reductions[s] = reduce(pre(data[s*L .. min((s+1)*L, n))))
//   scan_using_segment_reductions   results[s*L + i] = reduce(segment_reductions[0..s)) (+) scan of segment s up to i
//   scan_single_segment             one workgroup scans the whole input
//
// Same kernel template names, template parameters and parameter lists as the reference's
// src/kernels/reduction/scan.cuh:125-131, :406-417, :519-528 (+ launch parameters :201-250, :468-513,
// :542-576); same two-phase contract: the caller provides the scratch array of segment reductions, phase 2
// reduces the entries before its own segment itself (scan.cuh:420-443).  Differences by design:
//
//  * segments are taken at GRID STRIDE (the reference ties segment index to block index), so any grid is
//    correct; reduce_segments also writes the neutral value into entries [num_segments, gridDim.x), which
//    is what the reference's surplus blocks do (scan.cuh:155-165);
//  * execution is the Delta block path's (delta.hpp): a lane owns E = 16/sizeof(result) consecutive elements
//    per row, rows are scanned in registers + one 64-lane DPP scan with the operator's neutral value filling
//    the gaps, a wavefront takes R consecutive rows per sweep, carries cross wavefronts through a
//    double-buffered LDS table — one barrier per waves*R rows, 16-byte loads and stores;
//  * the operator must be associative AND commutative with a neutral value (all of cuda::functors' are).
//
// HBM bytes per element: phase 1 reads sizeof(In); phase 2 reads sizeof(In) and writes sizeof(result).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdexcept>

#include "giddy/common.hpp"
#include "giddy/functors/functors.hpp"
#include "giddy/kernels/launch_config_resolution_params.hpp"
#include "giddy/on_device/wave64.hpp"
#include "giddy/util/math.hpp"

namespace cuda {
namespace kernels {
namespace reduction {
namespace scan {

namespace inclusivity_t { enum : bool { Exclusive = false, Inclusive = true }; }

namespace detail {

enum : unsigned { rows_per_wave_per_sweep = 4, max_waves_per_block = 16, loads_in_flight = 4 };

template <typename Result> struct row_shape { enum : unsigned { E = sizeof(Result) >= 16 ? 1 : 16 / sizeof(Result), row = 64 * E }; };

/// every thread of the workgroup gets reduce(value over all threads); `table` holds max_waves_per_block entries
template <typename Op>
__device__ __forceinline__ typename Op::result_type reduce_across_block(typename Op::result_type value, typename Op::result_type* table)
{
	using T = typename Op::result_type;
	Op op;
	const T wave_result = wave64::read_last_lane(wave64::inclusive_scan<Op>(value));
	__syncthreads();   // (the table may still be read from an earlier use)
	if (wave64::lane_id() == 0) { table[wave64::index_in_block()] = wave_result; }
	__syncthreads();
	T all = Op::neutral_value();
	for (unsigned w = 0; w < wave64::per_block(); w++) { all = op(all, table[w]); }
	return all;
}

/// reduce(pre(data[start .. end))) by the whole workgroup (every thread gets it)
template <typename Op, typename In, typename Pre>
__device__ __forceinline__ typename Op::result_type reduce_range(const In* __restrict__ data, uint64_t start, uint64_t end, typename Op::result_type* table)
{
	using T = typename Op::result_type;
	constexpr unsigned E = row_shape<T>::E;
	constexpr unsigned in_bytes = E * sizeof(In) >= 16 ? 16 : E * sizeof(In);
	Op op; Pre pre;
	T mine = Op::neutral_value();
	// vector part: groups of E elements from the first group-aligned position on
	uint64_t p = start;
	const bool vectorizable = memory_access::is_aligned(data, in_bytes);
	if (vectorizable) {
		const uint64_t first_group = (start + E - 1) / E, end_group = end / E;
		if (first_group < end_group) {
			for (uint64_t q = start + threadIdx.x; q < first_group * E; q += blockDim.x) { mine = op(mine, (T) pre(data[q])); }
			const uint64_t stride = blockDim.x;
			uint64_t g = first_group + threadIdx.x;
			for (; g + (loads_in_flight - 1) * stride < end_group; g += loads_in_flight * stride) {
				memory_access::group<In, E> in[loads_in_flight];
				#pragma unroll
				for (unsigned k = 0; k < loads_in_flight; k++) { in[k] = memory_access::load_group<In, E>(data + (g + k * stride) * E); }
				#pragma unroll
				for (unsigned k = 0; k < loads_in_flight; k++) {
					#pragma unroll
					for (unsigned j = 0; j < E; j++) { mine = op(mine, (T) pre(in[k][j])); }
				}
			}
			for (; g < end_group; g += stride) {
				memory_access::group<In, E> in = memory_access::load_group<In, E>(data + g * E);
				#pragma unroll
				for (unsigned j = 0; j < E; j++) { mine = op(mine, (T) pre(in[j])); }
			}
			p = end_group * E;
		}
	}
	for (uint64_t q = p + threadIdx.x; q < end; q += blockDim.x) { mine = op(mine, (T) pre(data[q])); }
	return reduce_across_block<Op>(mine, table);
}

/// results[start .. end) = seed (+) scan of pre(data[start .. end)), by the whole workgroup
template <typename Op, typename In, bool Inclusive, typename Pre>
__device__ __forceinline__ void scan_range(
	typename Op::result_type* __restrict__ results, const In* __restrict__ data, uint64_t start, uint64_t end,
	typename Op::result_type seed, typename Op::result_type (*totals)[max_waves_per_block], unsigned& buffer)
{
	using T = typename Op::result_type;
	constexpr unsigned E = row_shape<T>::E, row = row_shape<T>::row, R = rows_per_wave_per_sweep;
	constexpr unsigned in_bytes = E * sizeof(In) >= 16 ? 16 : E * sizeof(In);
	Op op; Pre pre;
	const unsigned lane = wave64::lane_id(), w = wave64::index_in_block(), num_waves = wave64::per_block();
	const T neutral = Op::neutral_value();
	// rows are laid out from `start`; vector access needs start-relative groups to be aligned in memory
	const bool vectorizable = memory_access::is_aligned(results + start, 16) && memory_access::is_aligned(data + start, in_bytes);
	T carry = seed;

	for (uint64_t sweep_start = start; sweep_start < end; sweep_start += (uint64_t) num_waves * R * row) {
		const uint64_t mine = sweep_start + (uint64_t) w * R * row;
		T inclusive[R][E];   // within the row, per owned element
		T before_lane[R];    // reduction of the row's elements owned by lower lanes
		T row_total[R];
		T my_total = neutral;
		memory_access::group<In, E> in[R];
		#pragma unroll
		for (unsigned k = 0; k < R; k++) {
			const uint64_t p = mine + (uint64_t) k * row + (uint64_t) lane * E;
			if (vectorizable && p + E <= end) { in[k] = memory_access::load_group<In, E>(data + p); }
			else {
				#pragma unroll
				for (unsigned j = 0; j < E; j++) { if (p + j < end) { in[k][j] = data[p + j]; } }
			}
		}
		#pragma unroll
		for (unsigned k = 0; k < R; k++) {
			const uint64_t p = mine + (uint64_t) k * row + (uint64_t) lane * E;
			T running = neutral;
			#pragma unroll
			for (unsigned j = 0; j < E; j++) {
				if (p + j < end) { running = op(running, (T) pre(in[k][j])); }
				inclusive[k][j] = running;
			}
			const T across = wave64::inclusive_scan<Op>(running);
			before_lane[k] = wave64::shift_up_by_one<T>(across, neutral);
			row_total[k] = wave64::read_last_lane(across);
			my_total = op(my_total, row_total[k]);
		}
		if (lane == 0) { totals[buffer][w] = my_total; }
		__syncthreads();
		T before = neutral, all = neutral;
		for (unsigned v = 0; v < num_waves; v++) {
			const T t = totals[buffer][v];
			if (v < w) { before = op(before, t); }
			all = op(all, t);
		}
		T running = op(carry, before);
		#pragma unroll
		for (unsigned k = 0; k < R; k++) {
			const uint64_t p = mine + (uint64_t) k * row + (uint64_t) lane * E;
			const T base = op(running, before_lane[k]);
			memory_access::group<T, E> out;
			#pragma unroll
			for (unsigned j = 0; j < E; j++) {
				out[j] = Inclusive ? op(base, inclusive[k][j]) : (j == 0 ? base : op(base, inclusive[k][j - 1]));
			}
			if (vectorizable && p + E <= end) { memory_access::store_group(results + p, out); }
			else {
				#pragma unroll
				for (unsigned j = 0; j < E; j++) { if (p + j < end) { results[p + j] = out[j]; } }
			}
			running = op(running, row_total[k]);
		}
		carry = op(carry, all);
		buffer ^= 1;   // the next sweep writes the other table; this one is rewritten only after another barrier
	}
}

inline void validate_segment_length(size_t segment_length)
{
	if (segment_length == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
}

} // namespace detail

namespace reduce_segments {

template <unsigned IndexSize, typename ReductionOp, typename InputDatum,
	typename PretransformOp = functors::identity<typename ReductionOp::first_argument_type>>
__global__ void reduce_segments(
	typename ReductionOp::result_type*  __restrict__  segment_reductions,
	const InputDatum*                   __restrict__  data,
	size_type_by_index_size<IndexSize>                total_length,
	size_type_by_index_size<IndexSize>                full_segment_length)
{
	using result_type = typename ReductionOp::result_type;
	static_assert(std::is_same<typename ReductionOp::first_argument_type, result_type>::value &&
		std::is_same<typename ReductionOp::second_argument_type, result_type>::value,
		"Only all-same-argument reduction operations are supported");   // as scan.cuh:144-148
	__shared__ result_type table[detail::max_waves_per_block];
	const uint64_t n = total_length, L = full_segment_length;
	if (L == 0) { return; }
	const uint64_t num_segments = (n + L - 1) / L;
	const uint64_t entries = num_segments > gridDim.x ? num_segments : gridDim.x;
	for (uint64_t s = blockIdx.x; s < entries; s += gridDim.x) {
		result_type r = ReductionOp::neutral_value();
		if (s < num_segments) {
			const uint64_t start = s * L, end = (start + L < n) ? start + L : n;
			r = detail::reduce_range<ReductionOp, InputDatum, PretransformOp>(data, start, end, table);
		}
		if (threadIdx.x == 0) { segment_reductions[s] = r; }
	}
}

template <unsigned IndexSize, typename ReductionOp, typename InputDatum,
	typename PretransformOp = functors::identity<typename ReductionOp::first_argument_type>>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;

	/// `segment_length` given: one workgroup per segment (grid stride beyond what keeps the GPU busy).  Not given
	/// (as the reference allows, scan.cuh:188-199): the resolver picks a grid for `length_` elements at thread
	/// resolution and the caller derives the segment length from it, e.g. with suggested_segment_length().
	launch_config_resolution_params_t(
		device::properties_t device_properties_, size_t length_, optional<size_t> segment_length = nullopt)
	: parent(device_properties_, device_function_t(reduce_segments<IndexSize, ReductionOp, InputDatum, PretransformOp>).attributes())
	{
		if (segment_length) {
			detail::validate_segment_length(segment_length.value());
			grid_construction_resolution = block;
			length = util::div_rounding_up(length_, segment_length.value());
		}
		else {
			grid_construction_resolution = thread;
			length = util::div_rounding_up(length_, (size_t) detail::row_shape<typename ReductionOp::result_type>::E * detail::loads_in_flight);
		}
		serialization_option          = auto_maximized;
		quanta.threads_in_block       = warp_size;
		upper_limits.threads_in_block = 512;
		keep_gpu_busy_factor          = 8;
	}
};

/// A segment length that gives every workgroup of a `num_blocks`-workgroup grid one segment of whole sweeps
inline size_t suggested_segment_length(size_t length, size_t num_blocks, size_t threads_per_block, size_t result_size)
{
	size_t sweep = threads_per_block * detail::rows_per_wave_per_sweep * (result_size >= 16 ? 1 : 16 / result_size);
	size_t per_block = util::div_rounding_up(length, std::max<size_t>(1, num_blocks));
	return std::max<size_t>(sweep, util::div_rounding_up(per_block, sweep) * sweep);
}

} // namespace reduce_segments

namespace scan_using_segment_reductions {

template <unsigned IndexSize, typename ReductionOp, typename InputDatum, bool Inclusivity = inclusivity_t::Inclusive,
	typename PretransformOp = functors::identity<typename ReductionOp::first_argument_type>>
__global__ void scan_using_segment_reductions(
	typename ReductionOp::result_type*         __restrict__  results,
	const typename ReductionOp::result_type*   __restrict__  segment_reductions,
	const InputDatum*                          __restrict__  data,
	size_type_by_index_size<IndexSize>                       total_length,
	size_type_by_index_size<IndexSize>                       segment_length)
{
	using result_type = typename ReductionOp::result_type;
	__shared__ result_type table[detail::max_waves_per_block];
	__shared__ result_type totals[2][detail::max_waves_per_block];
	const uint64_t n = total_length, L = segment_length;
	if (L == 0 || wave64::per_block() > detail::max_waves_per_block) { return; }
	const uint64_t num_segments = (n + L - 1) / L;
	unsigned buffer = 0;
	for (uint64_t s = blockIdx.x; s < num_segments; s += gridDim.x) {
		// the reduction of everything before this segment, from the first phase's results (scan.cuh:420-443)
		const result_type before = detail::reduce_range<ReductionOp, result_type, functors::identity<result_type>>(
			segment_reductions, 0, s, table);
		const uint64_t start = s * L, end = (start + L < n) ? start + L : n;
		detail::scan_range<ReductionOp, InputDatum, Inclusivity, PretransformOp>(results, data, start, end, before, totals, buffer);
		__syncthreads();
	}
}

template <unsigned IndexSize, typename ReductionOp, typename InputDatum, bool Inclusivity = inclusivity_t::Inclusive,
	typename PretransformOp = functors::identity<typename ReductionOp::first_argument_type>>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;

	launch_config_resolution_params_t(
		device::properties_t device_properties_, size_t length_, size_t segment_length,
		optional<shared_memory_size_t> dynamic_shared_mem_limit = nullopt)
	: parent(device_properties_,
		device_function_t(scan_using_segment_reductions<IndexSize, ReductionOp, InputDatum, Inclusivity, PretransformOp>).attributes(),
		dynamic_shared_mem_limit)
	{
		detail::validate_segment_length(segment_length);
		grid_construction_resolution  = block;   // one workgroup per segment, segments at grid stride
		length                        = util::div_rounding_up(length_, segment_length);
		serialization_option          = auto_maximized;
		quanta.threads_in_block       = warp_size;
		// as many wavefronts as one sweep of the segment needs, at most 8
		size_t rows = util::div_rounding_up(segment_length, (size_t) detail::row_shape<typename ReductionOp::result_type>::row);
		size_t waves = std::min<size_t>(8, std::max<size_t>(1, util::div_rounding_up(rows, (size_t) detail::rows_per_wave_per_sweep)));
		block_resolution_constraints.max_threads_per_block = (grid_block_dimension_t) (waves * warp_size);
		upper_limits.threads_in_block = (grid_block_dimension_t) (waves * warp_size);
		keep_gpu_busy_factor          = 8;
	}
};

} // namespace scan_using_segment_reductions

namespace scan_single_segment {

template <unsigned IndexSize, typename ReductionOp, typename InputDatum, bool Inclusivity = inclusivity_t::Inclusive,
	typename PretransformOp = functors::identity<typename ReductionOp::first_argument_type>>
__global__ void scan_single_segment(
	typename ReductionOp::result_type*  __restrict__  result,
	const InputDatum*                   __restrict__  data,
	size_type_by_index_size<IndexSize>                length)
{
	using result_type = typename ReductionOp::result_type;
	__shared__ result_type totals[2][detail::max_waves_per_block];
	if (blockIdx.x != 0 || wave64::per_block() > detail::max_waves_per_block) { return; }   // one workgroup scans everything
	unsigned buffer = 0;
	detail::scan_range<ReductionOp, InputDatum, Inclusivity, PretransformOp>(
		result, data, 0, length, ReductionOp::neutral_value(), totals, buffer);
}

template <unsigned IndexSize, typename ReductionOp, typename InputDatum, bool Inclusivity = inclusivity_t::Inclusive,
	typename PretransformOp = functors::identity<typename ReductionOp::first_argument_type>>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;

	launch_config_resolution_params_t(
		device::properties_t device_properties_, size_t length_, optional<shared_memory_size_t> dynamic_shared_mem_limit = nullopt)
	: parent(device_properties_,
		device_function_t(scan_single_segment<IndexSize, ReductionOp, InputDatum, Inclusivity, PretransformOp>).attributes(),
		dynamic_shared_mem_limit)
	{
		grid_construction_resolution  = block;
		length                        = 1;
		serialization_option          = none;
		quanta.threads_in_block       = warp_size;
		size_t rows = util::div_rounding_up(std::max<size_t>(1, length_), (size_t) detail::row_shape<typename ReductionOp::result_type>::row);
		size_t waves = std::min<size_t>(detail::max_waves_per_block, std::max<size_t>(1, util::div_rounding_up(rows, (size_t) detail::rows_per_wave_per_sweep)));
		block_resolution_constraints.max_threads_per_block = (grid_block_dimension_t) (waves * warp_size);
		upper_limits.threads_in_block = (grid_block_dimension_t) (waves * warp_size);
	}
};

} // namespace scan_single_segment

} // namespace scan
} // namespace reduction
} // namespace kernels
} // namespace cuda
