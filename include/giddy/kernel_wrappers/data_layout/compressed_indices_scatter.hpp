// Kernel wrapper for the compressed-indices scatter.  Argument-map keys and exact types follow the
// reference's src/kernel_wrappers/data_layout/compressed_indices_scatter.cu:48-104 (an empty grid
// is a no-op, :69-73).
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/data_layout/compressed_indices_scatter.hpp"

namespace cuda {
namespace kernels {
namespace scatter {
namespace compressed_indices {

template <unsigned OutputIndexSize, unsigned ElementSize, unsigned InputIndexSize, unsigned RunLengthSize = InputIndexSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);

	using element_type      = uint_t<ElementSize>;
	using input_index_type  = uint_t<InputIndexSize>;
	using input_size_type   = size_type_by_index_size<InputIndexSize>;
	using output_index_type = uint_t<OutputIndexSize>;
	using output_size_type  = size_type_by_index_size<OutputIndexSize>;
	using run_length_type   = uint_t<RunLengthSize>;

	launch_configuration_t resolve_launch_configuration(
		device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
		size_t input_data_length, size_t anchoring_period, launch_configuration_limits_t limits) const
	{
		launch_config_resolution_params_t<OutputIndexSize, ElementSize, InputIndexSize, RunLengthSize> params(
			device_properties, kernel_function_attributes, input_data_length, anchoring_period);
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}
};

#define GIDDY_CIS_TEMPLATE template <unsigned OutputIndexSize, unsigned ElementSize, unsigned InputIndexSize, unsigned RunLengthSize>
#define GIDDY_CIS_KERNEL kernel_t<OutputIndexSize, ElementSize, InputIndexSize, RunLengthSize>

GIDDY_CIS_TEMPLATE
launch_configuration_t GIDDY_CIS_KERNEL::resolve_launch_configuration(
	device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
	arguments_type extra_arguments, launch_configuration_limits_t limits) const
{
	auto input_data_length = argument<size_t>(extra_arguments, "input_data_length");
	auto anchoring_period  = argument<size_t>(extra_arguments, "anchoring_period");
	return resolve_launch_configuration(device_properties, kernel_function_attributes, input_data_length, anchoring_period, limits);
}

GIDDY_CIS_TEMPLATE
void GIDDY_CIS_KERNEL::enqueue_launch_with(stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	if (launch_config.grid_dimensions == 0) { return; }   // no patches, nothing to do

	auto target                                       = argument<element_type*>(arguments, "target");
	auto data_to_scatter                              = argument<const element_type*>(arguments, "data_to_scatter");
	auto scatter_position_run_lengths                 = argument<const run_length_type*>(arguments, "scatter_position_run_lengths");
	auto scatter_position_run_individual_offset_sizes = argument<const unsigned char*>(arguments, "scatter_position_run_individual_offset_sizes");
	auto scatter_position_run_baseline_values         = argument<const output_index_type*>(arguments, "scatter_position_run_baseline_values");
	auto scatter_position_run_offsets_start_pos       = argument<const input_index_type*>(arguments, "scatter_position_run_offsets_start_pos");
	auto scatter_position_offset_bytes                = argument<const unsigned char*>(arguments, "scatter_position_offset_bytes");
	auto scatter_position_anchors                     = argument<const input_index_type*>(arguments, "scatter_position_anchors");
	auto anchoring_period                             = argument<input_index_type>(arguments, "anchoring_period");
	auto num_scatter_position_runs                    = argument<input_index_type>(arguments, "num_scatter_position_runs");
	auto input_data_length                            = argument<input_size_type>(arguments, "input_data_length");

	cuda::kernel::enqueue_launch(*this, stream, launch_config,
		target, data_to_scatter, scatter_position_run_lengths, scatter_position_run_individual_offset_sizes,
		scatter_position_run_baseline_values, scatter_position_run_offsets_start_pos, scatter_position_offset_bytes,
		scatter_position_anchors, anchoring_period, num_scatter_position_runs, input_data_length);
}

GIDDY_CIS_TEMPLATE
const device_function_t GIDDY_CIS_KERNEL::get_device_function() const
{
	return reinterpret_cast<const void*>(scatter<OutputIndexSize, ElementSize, InputIndexSize, RunLengthSize>);
}

#undef GIDDY_CIS_TEMPLATE
#undef GIDDY_CIS_KERNEL

} // namespace compressed_indices
} // namespace scatter
} // namespace kernels
} // namespace cuda
