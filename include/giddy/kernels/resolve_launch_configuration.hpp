// cuda::kernels::resolve_launch_configuration(params, limits, serialization_factor)
// (reference: src/kernels/resolve_launch_configuration.h:22-25).  Host-only; defined in
// libgiddy_amd/csrc/resolve_launch_configuration.cpp.
#pragma once
#include "giddy/kernel_wrapper.hpp"
#include "giddy/kernels/launch_config_resolution_params.hpp"

namespace cuda {
namespace kernels {

launch_configuration_t resolve_launch_configuration(
	const launch_config_resolution_params_t&  params,
	const launch_configuration_limits_t&      limits = { nullopt, nullopt, nullopt },
	optional<serialization_factor_t>          serialization_factor = nullopt);

} // namespace kernels
} // namespace cuda
