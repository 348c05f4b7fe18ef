// Kernel wrapper for the naive scatter (patch application).  Argument-map keys and exact types
// follow the reference's src/kernel_wrappers/data_layout/scatter.cu:49-78.
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/data_layout/scatter.hpp"

namespace cuda {
namespace kernels {
namespace scatter {

template <unsigned OutputIndexSize, unsigned ElementSize, unsigned InputIndexSize,
	serialization_factor_t SerializationFactor = DefaultSerializationFactor>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);

	using element_type      = uint_t<ElementSize>;
	using input_index_type  = uint_t<InputIndexSize>;
	using input_size_type   = size_type_by_index_size<InputIndexSize>;
	using output_index_type = uint_t<OutputIndexSize>;
	using output_size_type  = size_type_by_index_size<OutputIndexSize>;

	launch_configuration_t resolve_launch_configuration(
		device::properties_t           device_properties,
		device_function::attributes_t  kernel_function_attributes,
		size_t                         data_length,
		launch_configuration_limits_t  limits) const
	{
		launch_config_resolution_params_t<OutputIndexSize, ElementSize, InputIndexSize, SerializationFactor> params(
			device_properties, kernel_function_attributes, data_length);
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}
};

template <unsigned OutputIndexSize, unsigned ElementSize, unsigned InputIndexSize, serialization_factor_t SerializationFactor>
launch_configuration_t kernel_t<OutputIndexSize, ElementSize, InputIndexSize, SerializationFactor>::resolve_launch_configuration(
	device::properties_t            device_properties,
	device_function::attributes_t   kernel_function_attributes,
	arguments_type                  extra_arguments,
	launch_configuration_limits_t   limits) const
{
	auto data_length = argument<size_t>(extra_arguments, "data_length");
	return resolve_launch_configuration(device_properties, kernel_function_attributes, data_length, limits);
}

template <unsigned OutputIndexSize, unsigned ElementSize, unsigned InputIndexSize, serialization_factor_t SerializationFactor>
void kernel_t<OutputIndexSize, ElementSize, InputIndexSize, SerializationFactor>::enqueue_launch_with(
	stream::id_t                   stream,
	const launch_configuration_t&  launch_config,
	const arguments_type&                 arguments) const
{
	auto target       = argument<element_type*>(arguments, "target");
	auto data         = argument<const element_type*>(arguments, "data");
	auto indices      = argument<const output_index_type*>(arguments, "indices");
	auto data_length  = argument<input_size_type>(arguments, "data_length");

	cuda::kernel::enqueue_launch(*this, stream, launch_config, target, data, indices, data_length);
}

template <unsigned OutputIndexSize, unsigned ElementSize, unsigned InputIndexSize, serialization_factor_t SerializationFactor>
const device_function_t kernel_t<OutputIndexSize, ElementSize, InputIndexSize, SerializationFactor>::get_device_function() const
{
	return reinterpret_cast<const void*>(scatter<OutputIndexSize, ElementSize, InputIndexSize, SerializationFactor>);
}

} // namespace scatter
} // namespace kernels
} // namespace cuda
