// Launch-configuration resolution: declarative per-kernel parameters -> {grid, block, dynamic LDS}.
//
// Behaviour contract (what callers can observe) follows the reference's
// src/kernels/resolve_launch_configuration.cpp:114-409:
//   effective length  = length / serialization factor                         (:345-377)
//   threads per block = min over {device, LDS budget, kernel attribute, what the input needs,
//                       one-time limits, the parameters' thread / wavefront caps}, optionally
//                       rounded down to a power of two                          (:136-227)
//   block length      = the fixed length, or the maximum rounded down to the quantum (:229-260)
//   grid length       = ceil(effective length / units per block); auto-maximised grids are capped
//                       at  CUs x SIMDs-per-CU x keep_gpu_busy_factor / wavefronts-per-block (:262-332)
//   dynamic LDS       = per_block + threads x per_length_unit                   (:334-343)
// and throws the same exception types in the same situations (std::invalid_argument when a fixed
// block length cannot be honoured, when no thread fits, when a grid limit is exceeded;
// std::logic_error for serialization factors a kernel cannot accept).  The numbers it works with
// are CDNA4's: the "warp" quantum is a 64-lane wavefront and a "processor" is a compute unit.
#include "giddy/kernels/resolve_launch_configuration.hpp"

#include <algorithm>
#include <limits>
#include <stdexcept>
#include <string>

#include "giddy/util/math.hpp"

namespace cuda {
namespace kernels {

namespace {

using params_t = launch_config_resolution_params_t;
using limits_t = launch_configuration_limits_t;
constexpr size_t unconstrained = std::numeric_limits<size_t>::max();

size_t effective_length_of(const params_t& p, optional<serialization_factor_t> requested)
{
	switch (p.serialization_option) {
	case params_t::auto_maximized:
		return p.length;
	case params_t::none:
		if (requested.value_or(1) != 1) { throw std::logic_error("kernel does not allow for serialization"); }
		return p.length;
	case params_t::fixed_factor:
		if (requested.value_or(p.default_serialization_factor) != p.default_serialization_factor) {
			throw std::logic_error("kernel cannot admit serialization at a factor other than the fixed default");
		}
		break;
	case params_t::runtime_specified_factor:
		break;
	default:
		throw std::invalid_argument("Unsupported serialization option");
	}
	size_t factor = requested.value_or(p.default_serialization_factor);
	if (factor == 0) { throw std::invalid_argument("A serialization factor must be positive"); }
	return p.length == 0 ? 0 : 1 + (p.length - 1) / factor;
}

size_t block_length_quantum(const params_t& p)
{
	if (p.grid_construction_resolution == params_t::block && p.block_resolution_constraints.fixed_threads_per_block) {
		return p.block_resolution_constraints.fixed_threads_per_block.value();
	}
	size_t quantum = p.quanta.threads_in_block;
	if (p.quanta.warps_in_block) { quantum = util::lcm<size_t>(quantum, (size_t) p.quanta.warps_in_block.value() * warp_size); }
	if (p.grid_construction_resolution == params_t::warp) { quantum = util::lcm<size_t>(quantum, warp_size); }
	return quantum;
}

size_t max_threads_per_block(const params_t& p, const limits_t& limits, size_t effective_length)
{
	const size_t by_device = p.device_properties.maxThreadsDim[0];

	size_t threads_per_unit;
	switch (p.grid_construction_resolution) {
	case params_t::thread: threads_per_unit = 1; break;
	case params_t::warp:   threads_per_unit = warp_size; break;
	case params_t::block:  threads_per_unit = p.block_resolution_constraints.max_threads_per_block.value_or(by_device); break;
	default: throw std::invalid_argument("Unsupported grid construction resolution");
	}
	// no point in more threads than the whole input can occupy (rounded up to the quantum)
	const size_t by_input = util::round_up(effective_length * threads_per_unit, block_length_quantum(p));

	size_t by_lds = unconstrained;
	if (p.use_dynamic_shared_memory()) {
		auto per_block = p.dynamic_shared_memory_requirement.per_block.value_or(0);
		auto per_unit = p.dynamic_shared_memory_requirement.per_length_unit.value_or(0);
		if (per_block > p.available_dynamic_shared_memory_per_block) { by_lds = 0; }
		else if (per_unit != 0) { by_lds = (p.available_dynamic_shared_memory_per_block - per_block) / per_unit; }
	}

	size_t by_kernel = by_device;
	if (p.grid_construction_resolution == params_t::thread) { by_kernel = p.kernel_function_attributes.maxThreadsPerBlock; }
	else if (p.grid_construction_resolution == params_t::warp) {
		by_kernel = util::round_down<size_t>(p.kernel_function_attributes.maxThreadsPerBlock, warp_size);
	}

	size_t result = std::min({ by_device, by_input, by_lds, by_kernel,
		(size_t) limits.block_size.value_or(unconstrained),
		(size_t) p.upper_limits.threads_in_block.value_or(unconstrained),
		p.upper_limits.warps_in_block ? (size_t) p.upper_limits.warps_in_block.value() * warp_size : unconstrained });
	if (p.must_have_power_of_2_threads_per_block) { result = util::round_down_to_power_of_2<size_t>(result); }
	return result;
}

size_t block_length_of(const params_t& p, size_t max_threads)
{
	if (p.grid_construction_resolution == params_t::block && p.block_resolution_constraints.fixed_threads_per_block) {
		auto fixed = p.block_resolution_constraints.fixed_threads_per_block.value();
		if (fixed > max_threads) {
			throw std::invalid_argument("Kernel requires " + std::to_string(fixed) + " threads per block, while the different "
				"constraints on the number of threads per block only allow " + std::to_string(max_threads) + ".");
		}
		return fixed;
	}
	size_t length = util::round_down(max_threads, block_length_quantum(p));
	if (length == 0) {
		throw std::invalid_argument("The combined constraints on the number of threads per block do not allow any threads in a block");
	}
	return length;
}

size_t blocks_to_keep_gpu_busy(const params_t& p, size_t threads_per_block)
{
	size_t waves_per_block = util::div_rounding_up(threads_per_block, (size_t) warp_size);
	return std::max<size_t>(1,
		(size_t) p.device_properties.multiProcessorCount
		* p.device_properties.compute_capability().max_warp_schedulings_per_processor_cycle()
		* p.keep_gpu_busy_factor / waves_per_block);
}

size_t grid_length_of(const params_t& p, size_t effective_length, size_t threads_per_block, optional<grid_dimension_t> grid_limit)
{
	size_t units_per_block;
	switch (p.grid_construction_resolution) {
	case params_t::thread: units_per_block = threads_per_block; break;
	case params_t::warp:   units_per_block = threads_per_block / warp_size; break;
	case params_t::block:  units_per_block = 1; break;
	default: throw std::invalid_argument("Unsupported grid construction resolution");
	}
	size_t covering = util::div_rounding_up(effective_length, units_per_block);
	size_t limit = grid_limit ? (size_t) grid_limit.value() : covering;
	if (p.serialization_option == params_t::auto_maximized) {
		return std::min({ covering, blocks_to_keep_gpu_busy(p, threads_per_block), limit });
	}
	if (covering > limit) {
		throw std::invalid_argument("Required grid length exceeds specified limit: "
			+ std::to_string(limit) + " > " + std::to_string(covering));
	}
	return covering;
}

} // namespace

launch_configuration_t resolve_launch_configuration(
	const launch_config_resolution_params_t& params,
	const launch_configuration_limits_t& limits,
	optional<serialization_factor_t> serialization_factor)
{
	size_t effective_length = effective_length_of(params, serialization_factor);
	if (effective_length == 0) {
		// Empty input: nothing to launch.  (The reference divides by zero here; an empty grid is
		// the useful answer, and cuda::enqueue_launch treats it as a no-op.)
		linear_grid::launch_configuration_t empty;
		empty.grid_length = 0;
		empty.block_length = (grid_block_dimension_t) std::max<size_t>(block_length_quantum(params), 1);
		empty.dynamic_shared_memory_size = 0;
		return empty;
	}
	size_t max_threads = max_threads_per_block(params, limits, effective_length);
	size_t block_length = block_length_of(params, max_threads);
	size_t grid_length = grid_length_of(params, effective_length, block_length, limits.grid_size);
	if (grid_length > (size_t) std::numeric_limits<int>::max()) {
		throw std::invalid_argument("Required grid length exceeds what the device can schedule: " + std::to_string(grid_length));
	}
	size_t dynamic_lds =
		params.dynamic_shared_memory_requirement.per_block.value_or(0) +
		block_length * params.dynamic_shared_memory_requirement.per_length_unit.value_or(0);

	if (params.resident_blocks_per_multiprocessor_cap && params.resident_blocks_per_multiprocessor_cap.value() > 0) {
		// reserve so much LDS that exactly `cap` workgroups fit a compute unit.  (Not all of a CU's 160 KiB can be had:
		// measured, five 32 KiB or three 53 KiB workgroups do not fit, three of 52 KiB do — a 4 KiB margin.)
		size_t lds_per_cu = params.device_properties.maxSharedMemoryPerMultiProcessor ? params.device_properties.maxSharedMemoryPerMultiProcessor : 65536;
		size_t per_block = ((lds_per_cu - 4096) / params.resident_blocks_per_multiprocessor_cap.value()) / 1024 * 1024;
		size_t statically_used = params.kernel_function_attributes.sharedSizeBytes;
		size_t reserve = per_block > statically_used ? per_block - statically_used : 0;
		reserve = std::min<size_t>(reserve, params.available_dynamic_shared_memory_per_block);
		if (limits.dynamic_shared_memory) { reserve = std::min<size_t>(reserve, limits.dynamic_shared_memory.value()); }
		dynamic_lds = std::max(dynamic_lds, reserve);
	}

	linear_grid::launch_configuration_t config;
	config.grid_length = (grid_dimension_t) grid_length;
	config.block_length = (grid_block_dimension_t) block_length;
	config.dynamic_shared_memory_size = (shared_memory_size_t) dynamic_lds;
	return config;
}

} // namespace kernels
} // namespace cuda
