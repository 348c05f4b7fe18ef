// Kernel wrapper for Delta decompression.  Argument-map keys and exact types follow the
// reference's src/kernel_wrappers/decompression/delta.cu:49-83.
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/decompression/delta.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace delta {

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);

	using compressed_type   = uint_t<CompressedSize>;
	using uncompressed_type = uint_t<UncompressedSize>;
	using size_type         = size_type_by_index_size<IndexSize>;

	launch_configuration_t resolve_launch_configuration(
		device::properties_t           device_properties,
		device_function::attributes_t  kernel_function_attributes,
		size_t                         length,
		size_t                         segment_length,
		launch_configuration_limits_t  limits) const
	{
		launch_config_resolution_params_t<IndexSize, UncompressedSize, CompressedSize> params(
			device_properties, kernel_function_attributes, length, segment_length, limits.dynamic_shared_memory);
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}
};

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
launch_configuration_t kernel_t<IndexSize, UncompressedSize, CompressedSize>::resolve_launch_configuration(
	device::properties_t           device_properties,
	device_function::attributes_t  kernel_function_attributes,
	arguments_type                 extra_arguments,
	launch_configuration_limits_t  limits) const
{
	auto length         = argument<size_t>(extra_arguments, "length");
	auto segment_length = argument<size_t>(extra_arguments, "segment_length");
	return resolve_launch_configuration(device_properties, kernel_function_attributes, length, segment_length, limits);
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
void kernel_t<IndexSize, UncompressedSize, CompressedSize>::enqueue_launch(
	stream::id_t                   stream,
	const launch_configuration_t&  launch_config,
	arguments_type                 arguments) const
{
	auto decompressed      = argument<uncompressed_type*>(arguments, "decompressed");
	auto compressed_input  = argument<const compressed_type*>(arguments, "compressed_input");
	auto anchor_values     = argument<const uncompressed_type*>(arguments, "anchor_values");
	auto length            = argument<uint_t<IndexSize>>(arguments, "length");
	auto segment_length    = argument<uint_t<IndexSize>>(arguments, "segment_length");

	cuda::kernel::enqueue_launch(*this, stream, launch_config,
		decompressed, compressed_input, anchor_values, (size_type) length, (size_type) segment_length);
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
const device_function_t kernel_t<IndexSize, UncompressedSize, CompressedSize>::get_device_function() const
{
	return reinterpret_cast<const void*>(decompress<IndexSize, UncompressedSize, CompressedSize>);
}

} // namespace delta
} // namespace decompression
} // namespace kernels
} // namespace cuda
