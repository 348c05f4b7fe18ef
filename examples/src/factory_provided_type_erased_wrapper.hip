// Mode of use A: a type-erased wrapper produced by the factory from its key string; nothing about the
// kernel is known at compile time (cf. the reference's README.md:117-126).
#include "common.hpp"

#include "kernel_wrappers/registered_wrapper.h"

void decompress_on_device(
	uncompressed_type* decompressed, const compressed_type* compressed,
	const model_coefficients_type* segment_model_coefficients, size_type length, size_type segment_length)
{
	auto kernel_name =
		"decompression::frame_of_reference::kernel_t<4u, int, short, cuda::functors::unary::parametric_model::constant<4u, int> >";
	auto type_erased_kernel = cuda::registered::kernel_t::produceSubclass(kernel_name);

	auto device_properties = cuda::device::current::get().properties();
	auto kernel_function_attributes = type_erased_kernel->get_device_function().attributes();

	cuda::registered::kernel_t::arguments_type config_resolution_arguments = {
		{ "length",         (size_t) length         },
		{ "segment_length", (size_t) segment_length },
	};
	auto launch_config = type_erased_kernel->resolve_launch_configuration(
		device_properties, kernel_function_attributes, config_resolution_arguments);

	cuda::registered::kernel_t::arguments_type launch_arguments = {
		{ "decompressed",               decompressed                },
		{ "compressed_input",           compressed                  },
		{ "segment_model_coefficients", segment_model_coefficients  },
		{ "length",                     length                      },
		{ "segment_length",             segment_length              },
	};
	type_erased_kernel->enqueue_launch(cuda::stream::default_stream_id, launch_config, launch_arguments);
}
