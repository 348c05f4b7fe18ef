// Kernel wrapper for Model decompression.  Argument-map keys and exact types follow the reference's
// src/kernel_wrappers/decompression/model.cu:44-75 ("model_coefficients" is a coefficients_type BY VALUE).
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/decompression/model.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace model {

template <unsigned IndexSize, typename Uncompressed, typename UnaryModelFunction>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);

	using model_coefficients = typename UnaryModelFunction::coefficients_type;
	using size_type          = size_type_by_index_size<IndexSize>;

	launch_configuration_t resolve_launch_configuration(
		device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
		size_t length, launch_configuration_limits_t limits) const
	{
		launch_config_resolution_params_t<IndexSize, Uncompressed, UnaryModelFunction> params(
			device_properties, kernel_function_attributes, length);
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}
};

template <unsigned IndexSize, typename Uncompressed, typename UnaryModelFunction>
launch_configuration_t kernel_t<IndexSize, Uncompressed, UnaryModelFunction>::resolve_launch_configuration(
	device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
	arguments_type extra_arguments, launch_configuration_limits_t limits) const
{
	auto length = argument<size_t>(extra_arguments, "length");
	return resolve_launch_configuration(device_properties, kernel_function_attributes, length, limits);
}

template <unsigned IndexSize, typename Uncompressed, typename UnaryModelFunction>
void kernel_t<IndexSize, Uncompressed, UnaryModelFunction>::enqueue_launch_with(
	stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	using model_coefficients_type = typename UnaryModelFunction::coefficients_type;
	auto length             = argument<uint_t<IndexSize>>(arguments, "length");
	auto decompressed       = argument<Uncompressed*>(arguments, "decompressed");
	auto model_coefficients = argument<model_coefficients_type>(arguments, "model_coefficients");
	cuda::kernel::enqueue_launch(*this, stream, launch_config, decompressed, model_coefficients, (size_type) length);
}

template <unsigned IndexSize, typename Uncompressed, typename UnaryModelFunction>
const device_function_t kernel_t<IndexSize, Uncompressed, UnaryModelFunction>::get_device_function() const
{
	return reinterpret_cast<const void*>(decompress<IndexSize, Uncompressed, UnaryModelFunction>);
}

} // namespace model
} // namespace decompression
} // namespace kernels
} // namespace cuda
