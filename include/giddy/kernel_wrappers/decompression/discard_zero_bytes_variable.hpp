// Kernel wrapper for variable-width discard-zero-bytes decompression.  Argument-map keys and exact
// types follow the reference's src/kernel_wrappers/decompression/discard_zero_bytes_variable.cu:52-95.
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/decompression/discard_zero_bytes_variable.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace discard_zero_bytes {
namespace variable_width {

template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize = sizeof(native_word_t)>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);

	using element_size_t     = unsigned;
	using uncompressed_type  = uint_t<UncompressedSize>;
	using element_sizes_container_size_type = uint_t<ElementSizesContainerSize>;
	using size_type          = size_type_by_index_size<IndexSize>;

	launch_configuration_t resolve_launch_configuration(
		device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
		size_t length_in_elements, size_t segment_length, launch_configuration_limits_t limits) const
	{
		launch_config_resolution_params_t<IndexSize, UncompressedSize, ElementSizesContainerSize> params(
			device_properties, kernel_function_attributes, length_in_elements, segment_length);
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}
};

template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
launch_configuration_t kernel_t<IndexSize, UncompressedSize, ElementSizesContainerSize>::resolve_launch_configuration(
	device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
	arguments_type extra_arguments, launch_configuration_limits_t limits) const
{
	auto length         = argument<size_t>(extra_arguments, "length");          // overall, in elements
	auto segment_length = argument<size_t>(extra_arguments, "segment_length");  // elements per anchored segment
	return resolve_launch_configuration(device_properties, kernel_function_attributes, length, segment_length, limits);
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
void kernel_t<IndexSize, UncompressedSize, ElementSizesContainerSize>::enqueue_launch(
	stream::id_t stream, const launch_configuration_t& launch_config, arguments_type arguments) const
{
	using index_type = uint_t<IndexSize>;
	auto decompressed                 = argument<uncompressed_type*>(arguments, "decompressed");
	auto compressed_data              = argument<const unsigned char*>(arguments, "compressed_data");
	auto packed_element_sizes         = argument<const element_sizes_container_size_type*>(arguments, "packed_element_sizes");
	auto position_anchors             = argument<const index_type*>(arguments, "position_anchors");
	auto segment_length               = argument<index_type>(arguments, "segment_length");
	auto length                       = argument<index_type>(arguments, "length");
	auto min_represented_element_size = argument<element_size_t>(arguments, "min_represented_element_size");
	auto bits_per_element_size        = argument<unsigned>(arguments, "bits_per_element_size");

	cuda::kernel::enqueue_launch(*this, stream, launch_config,
		decompressed, compressed_data, packed_element_sizes, position_anchors,
		(size_type) segment_length, (size_type) length, min_represented_element_size, bits_per_element_size);
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
const device_function_t kernel_t<IndexSize, UncompressedSize, ElementSizesContainerSize>::get_device_function() const
{
	return reinterpret_cast<const void*>(decompress<IndexSize, UncompressedSize, ElementSizesContainerSize>);
}

} // namespace variable_width
} // namespace discard_zero_bytes
} // namespace decompression
} // namespace kernels
} // namespace cuda
