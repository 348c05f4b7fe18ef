// Mode of use B: instantiate the wrapper template directly and use its typed interface
// (cf. the reference's README.md:106-115).
#include "common.hpp"

#include "kernel_wrappers/decompression/frame_of_reference.cu"

void decompress_on_device(
	uncompressed_type* decompressed, const compressed_type* compressed,
	const model_coefficients_type* segment_model_coefficients, size_type length, size_type segment_length)
{
	namespace kernel_ns = cuda::kernels::decompression::frame_of_reference;
	kernel_ns::kernel_t<sizeof(index_type), uncompressed_type, compressed_type, model_type> kernel;

	auto device_properties = cuda::device::current::get().properties();
	auto kernel_function_attributes = kernel.get_device_function().attributes();
	auto launch_config = kernel.resolve_launch_configuration(
		device_properties, kernel_function_attributes, length, segment_length);

	index_type segments_per_block = 1;   // signature parity only: this library's kernel tiles independently of it
	cuda::kernel::enqueue_launch(kernel, launch_config,
		decompressed, compressed, segment_model_coefficients, length, segment_length, segments_per_block);
}
