// Declarative description of how a kernel wants to be launched; consumed by
// cuda::kernels::resolve_launch_configuration().  Field names, enumerators and defaults follow the
// reference's src/kernels/launch_config_resolution_params.h:12-134 — per-kernel parameter classes
// (in giddy/kernels/**) fill them in with numbers re-derived for CDNA4: 64-lane wavefronts as the
// `warp` quantum, 256 CUs x 4 SIMDs in the keep-busy rule, 64 KiB default dynamic LDS per workgroup.
#pragma once
#include "giddy/api/types.hpp"
#include "giddy/optional_and_any.hpp"

namespace cuda {
namespace kernels {

class launch_config_resolution_params_t {
public:
	/// Which part of the grid handles one unit of `length`
	enum resolution_t { thread, warp, block };

	enum serialization_option_t {
		none,                      ///< one length unit per thread / wavefront / workgroup
		fixed_factor,              ///< the kernel hard-codes how many units each one takes
		runtime_specified_factor,  ///< the kernel takes the factor as an argument
		auto_maximized,            ///< grid-stride: as few workgroups as keep the GPU busy
		keep_gpu_busy = auto_maximized,
	};

	device::properties_t           device_properties;
	device_function::attributes_t  kernel_function_attributes;
	size_t                         length { 0 };  ///< in units of the resolution
	resolution_t                   grid_construction_resolution { thread };
	serialization_option_t         serialization_option { none };
	serialization_factor_t         default_serialization_factor { 1 };

	struct {
		grid_block_dimension_t            threads_in_block { 1 };
		optional<grid_block_dimension_t>  warps_in_block   { nullopt };
		optional<grid_dimension_t>        blocks_in_grid   { nullopt };
	} quanta;

	struct {
		optional<grid_block_dimension_t>  threads_in_block { nullopt };
		optional<grid_block_dimension_t>  warps_in_block   { nullopt };
	} upper_limits;

	bool must_have_power_of_2_threads_per_block { false };

	struct {
		optional<shared_memory_size_t>  per_length_unit { nullopt };
		optional<shared_memory_size_t>  per_block       { nullopt };
	} dynamic_shared_memory_requirement;

	struct {
		optional<grid_block_dimension_t>  fixed_threads_per_block { nullopt };
		optional<grid_block_dimension_t>  max_threads_per_block   { nullopt };
	} block_resolution_constraints;

	shared_memory_size_t  available_dynamic_shared_memory_per_block;

	/// workgroups-per-SIMD multiplier for auto_maximized grids
	unsigned              keep_gpu_busy_factor { 50 };

	/// Occupancy throttle: at most this many workgroups resident on a compute unit, enforced the only way a launch
	/// can — by reserving dynamic LDS (160 KiB per CU on MI355X).  Write-heavy streams go FASTER with fewer
	/// resident workgroups (fewer write streams interleaved at the memory controllers); the per-kernel values
	/// were measured (tools/occupancy_sweep.py).  Unset: whatever fits.
	optional<unsigned>    resident_blocks_per_multiprocessor_cap { nullopt };

	bool use_dynamic_shared_memory() const
	{
		return dynamic_shared_memory_requirement.per_block || dynamic_shared_memory_requirement.per_length_unit;
	}

protected:
	launch_config_resolution_params_t(
		device::properties_t device_properties_,
		device_function::attributes_t kernel_function_attributes_,
		optional<shared_memory_size_t> shared_memory_size_limit = nullopt)
	: device_properties(device_properties_), kernel_function_attributes(kernel_function_attributes_)
	{
		available_dynamic_shared_memory_per_block =
			device_function::maximum_dynamic_shared_memory_per_block(kernel_function_attributes, device_properties);
		if (shared_memory_size_limit && shared_memory_size_limit.value() < available_dynamic_shared_memory_per_block) {
			available_dynamic_shared_memory_per_block = shared_memory_size_limit.value();
		}
	}

	launch_config_resolution_params_t(
		device::properties_t device_properties_,
		device_function_t device_function,
		optional<shared_memory_size_t> shared_memory_size_limit = nullopt)
	: launch_config_resolution_params_t(device_properties_, device_function.attributes(), shared_memory_size_limit) { }
};

} // namespace kernels
} // namespace cuda
