// Mode of use C: no wrapper at all — the declarative launch parameters, the resolver and the
// __global__ kernel template, directly (cf. the reference's README.md:94-104).
#include "common.hpp"

#include "kernels/resolve_launch_configuration.h"
#include "kernels/decompression/frame_of_reference.cuh"

void decompress_on_device(
	uncompressed_type* decompressed, const compressed_type* compressed,
	const model_coefficients_type* segment_model_coefficients, size_type length, size_type segment_length)
{
	namespace kernel_ns = cuda::kernels::decompression::frame_of_reference;

	auto device_properties = cuda::device::current::get().properties();
	kernel_ns::launch_config_resolution_params_t<sizeof(index_type), uncompressed_type, compressed_type, model_type>
		params(device_properties, length, segment_length);
	auto launch_config = cuda::kernels::resolve_launch_configuration(params);

	auto kernel_function = kernel_ns::decompress<sizeof(index_type), uncompressed_type, compressed_type, model_type>;
	auto num_blocks = launch_config.grid_dimensions.x;
	auto num_segments = util::div_rounding_up(length, segment_length);
	index_type segments_per_block = util::div_rounding_up(num_segments, num_blocks);

	cuda::launch(kernel_function, launch_config,
		decompressed, compressed, segment_model_coefficients, length, segment_length, segments_per_block);
}
