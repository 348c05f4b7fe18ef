// Kernel wrapper for incidence-bitmaps decompression.  Argument-map keys and exact types follow the
// reference's src/kernel_wrappers/decompression/incidence_bitmaps.cu:47-79.
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/decompression/incidence_bitmaps.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace incidence_bitmaps {

template <unsigned IndexSize, unsigned UncompressedSize, unsigned BitmapAlignmentInInts, bool UseDictionary = true>
class kernel_t : public cuda::registered::kernel_t {
public:
	using bit_container_t   = standard_bit_container_t;
	using uncompressed_type = uint_t<UncompressedSize>;
	using bitmap_index_type = unsigned char;
	using size_type         = size_type_by_index_size<IndexSize>;
	using num_bitmaps_type  = size_type_by_index_size<sizeof(bitmap_index_type)>;

	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);

	launch_configuration_t resolve_launch_configuration(
		device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
		size_t length, launch_configuration_limits_t limits) const
	{
		launch_config_resolution_params_t<IndexSize, UncompressedSize, BitmapAlignmentInInts, UseDictionary> params(
			device_properties, kernel_function_attributes, length);
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}
};

#define GIDDY_IB_TEMPLATE template <unsigned IndexSize, unsigned UncompressedSize, unsigned BitmapAlignmentInInts, bool UseDictionary>
#define GIDDY_IB_KERNEL kernel_t<IndexSize, UncompressedSize, BitmapAlignmentInInts, UseDictionary>

GIDDY_IB_TEMPLATE
launch_configuration_t GIDDY_IB_KERNEL::resolve_launch_configuration(
	device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
	arguments_type extra_arguments, launch_configuration_limits_t limits) const
{
	auto length = argument<size_t>(extra_arguments, "length");
	return resolve_launch_configuration(device_properties, kernel_function_attributes, length, limits);
}

GIDDY_IB_TEMPLATE
void GIDDY_IB_KERNEL::enqueue_launch_with(stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	auto decompressed       = argument<uncompressed_type*>(arguments, "decompressed");
	auto incidence_bitmaps  = argument<const bit_container_t*>(arguments, "incidence_bitmaps");
	auto dictionary_entries = argument<const uncompressed_type*>(arguments, "dictionary_entries");
	auto bitmap_length      = argument<size_type>(arguments, "bitmap_length");
	auto num_bitmaps        = argument<num_bitmaps_type>(arguments, "num_bitmaps");
	cuda::kernel::enqueue_launch(*this, stream, launch_config,
		decompressed, incidence_bitmaps, dictionary_entries, bitmap_length, num_bitmaps);
}

GIDDY_IB_TEMPLATE
const device_function_t GIDDY_IB_KERNEL::get_device_function() const
{
	return reinterpret_cast<const void*>(decompress<IndexSize, UncompressedSize, BitmapAlignmentInInts, UseDictionary>);
}

#undef GIDDY_IB_TEMPLATE
#undef GIDDY_IB_KERNEL

} // namespace incidence_bitmaps
} // namespace decompression
} // namespace kernels
} // namespace cuda
