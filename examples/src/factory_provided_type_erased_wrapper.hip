// Mode of use A: the kernel is known only by its factory key — a string — at run time.  The factory
// produces a type-erased wrapper; sizes go in as a name -> value map to resolve the launch
// configuration, buffers go in the same way to enqueue the launch (cf. the reference's README.md:117-126).
#include "common.hpp"

#include "kernel_wrappers/registered_wrapper.h"

namespace {

const char* const frame_of_reference_int32_from_int16 =
	"decompression::frame_of_reference::kernel_t<4u, int, short, "
	"cuda::functors::unary::parametric_model::constant<4u, int> >";

} // namespace

void decompress_on_device(
	uncompressed_type* decompressed, const compressed_type* compressed_input,
	const model_coefficients_type* segment_model_coefficients, size_type length, size_type segment_length)
{
	using wrapper = cuda::registered::kernel_t;
	if (!wrapper::canProduce(frame_of_reference_int32_from_int16)) {
		std::cerr << "the library does not register " << frame_of_reference_int32_from_int16 << "\n";
		std::exit(EXIT_FAILURE);
	}
	std::unique_ptr<wrapper> kernel = wrapper::produceSubclass(frame_of_reference_int32_from_int16);

	// launch-configuration resolution takes plain size_t's
	wrapper::arguments_type sizes;
	sizes["length"] = (size_t) length;
	sizes["segment_length"] = (size_t) segment_length;
	cuda::launch_configuration_t configuration = kernel->resolve_launch_configuration(
		cuda::device::current::get().properties(), kernel->get_device_function().attributes(), sizes);

	// the launch takes every argument in exactly the type the kernel declares; the macro names each
	// map entry after the variable it is built from
	wrapper::arguments_type launch_arguments =
		MAKE_KERNEL_ARGUMENTS(decompressed, compressed_input, segment_model_coefficients, length, segment_length);
	kernel->enqueue_launch(cuda::stream::default_stream_id, configuration, launch_arguments);
}
