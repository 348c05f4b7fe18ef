// Kernel wrapper for run-length decoding.  Argument-map keys and exact types follow the reference's
// src/kernel_wrappers/decompression/run_length_encoding.cu:56-93.
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/decompression/run_length_encoding.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace run_length_encoding {

template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize = UncompressedIndexSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);

	using run_index_type                = uint_t<RunIndexSize>;
	using uncompressed_index_type       = uint_t<UncompressedIndexSize>;
	using uncompressed_type             = uint_t<UncompressedSize>;
	using run_length_type               = uint_t<RunLengthSize>;
	using uncompressed_index_size_type  = size_type_by_index_size<UncompressedIndexSize>;
	using run_index_size_type           = size_type_by_index_size<RunIndexSize>;
	using size_type                     = uncompressed_index_size_type;

	static_assert(util::is_power_of_2(UncompressedIndexSize), "UncompressedIndexSize is not a power of 2");
	static_assert(util::is_power_of_2(RunIndexSize), "RunIndexSize is not a power of 2");
	static_assert(util::is_power_of_2(RunLengthSize), "RunLengthSize is not a power of 2");

	launch_configuration_t resolve_launch_configuration(
		device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
		size_t uncompressed_length, size_t segment_length, launch_configuration_limits_t limits) const
	{
		launch_config_resolution_params_t<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize> params(
			device_properties, kernel_function_attributes, uncompressed_length, segment_length);
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}
};

#define GIDDY_RLE_TEMPLATE template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize>
#define GIDDY_RLE_KERNEL kernel_t<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize>

GIDDY_RLE_TEMPLATE
launch_configuration_t GIDDY_RLE_KERNEL::resolve_launch_configuration(
	device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
	arguments_type extra_arguments, launch_configuration_limits_t limits) const
{
	auto uncompressed_length = argument<size_t>(extra_arguments, "length");
	auto segment_length      = argument<size_t>(extra_arguments, "segment_length");
	return resolve_launch_configuration(device_properties, kernel_function_attributes, uncompressed_length, segment_length, limits);
}

GIDDY_RLE_TEMPLATE
void GIDDY_RLE_KERNEL::enqueue_launch(stream::id_t stream, const launch_configuration_t& launch_config, arguments_type arguments) const
{
	auto decompressed             = argument<uncompressed_type*>(arguments, "decompressed");
	auto run_data                 = argument<const uncompressed_type*>(arguments, "run_data");
	auto run_lengths              = argument<const run_length_type*>(arguments, "run_lengths");
	auto position_anchors         = argument<const run_index_type*>(arguments, "position_anchors");
	auto intra_run_anchor_offsets = argument<const run_length_type*>(arguments, "intra_run_anchor_offsets");
	auto segment_length           = argument<size_type>(arguments, "segment_length");
	auto num_segments             = argument<size_type>(arguments, "num_segments");
	auto num_element_runs         = argument<run_index_size_type>(arguments, "num_element_runs");
	auto uncompressed_length      = argument<size_type>(arguments, "length");

	cuda::kernel::enqueue_launch(*this, stream, launch_config,
		decompressed, run_data, run_lengths, position_anchors, intra_run_anchor_offsets,
		segment_length, num_segments, num_element_runs, uncompressed_length);
}

GIDDY_RLE_TEMPLATE
const device_function_t GIDDY_RLE_KERNEL::get_device_function() const
{
	return reinterpret_cast<const void*>(decompress<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize>);
}

#undef GIDDY_RLE_TEMPLATE
#undef GIDDY_RLE_KERNEL

} // namespace run_length_encoding
} // namespace decompression
} // namespace kernels
} // namespace cuda
