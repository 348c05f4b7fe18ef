// Kernel wrapper for Frame-of-Reference decompression fused with patching (an extension, SURVEY.md
// §8f.1).  Argument-map keys: those of the FOR wrapper (src/kernel_wrappers/decompression/
// frame_of_reference.cu:58-102) plus "patch_values", "patch_positions" (ascending) and "num_patches" — or, in place
// of "patch_positions", the compressed-indices arrays and a "patch_positions_scratch" (patched/patch_positions.hpp).
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernel_wrappers/decompression/patched/patch_positions.hpp"
#include "giddy/kernel_wrappers/decompression/patched/decode_then_scatter.hpp"
#include "giddy/kernels/decompression/frame_of_reference.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace frame_of_reference {
namespace patched {

template <unsigned IndexSize, typename Uncompressed, typename Compressed, typename UnaryModelFunction>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);

	using index_type              = uint_t<IndexSize>;
	using size_type               = size_type_by_index_size<IndexSize>;
	using model_coefficients_type = typename UnaryModelFunction::coefficients_type;

	launch_configuration_t resolve_launch_configuration(
		device::properties_t           device_properties,
		device_function::attributes_t  kernel_function_attributes,
		size_t                         length,
		size_t                         segment_length,
		launch_configuration_limits_t  limits = { nullopt, nullopt, nullopt }) const
	{
		frame_of_reference::launch_config_resolution_params_t<IndexSize, Uncompressed, Compressed, UnaryModelFunction> params(
			device_properties, kernel_function_attributes, length, segment_length);
		// a thread of the patched kernel covers a wavefront chunk's share of elements rather than a tile's
		params.length = util::div_rounding_up(length, (size_t) stream_tiles::elements_per_thread_per_chunk<Uncompressed>());
		params.resident_blocks_per_multiprocessor_cap = nullopt;   // (the plain kernel's occupancy throttle was not measured for this one)
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}
};

template <unsigned IndexSize, typename Uncompressed, typename Compressed, typename UnaryModelFunction>
launch_configuration_t kernel_t<IndexSize, Uncompressed, Compressed, UnaryModelFunction>::resolve_launch_configuration(
	device::properties_t           device_properties,
	device_function::attributes_t  kernel_function_attributes,
	arguments_type                 extra_arguments,
	launch_configuration_limits_t  limits) const
{
	auto length         = argument<size_t>(extra_arguments, "length");
	auto segment_length = argument<size_t>(extra_arguments, "segment_length");
	return resolve_launch_configuration(device_properties, kernel_function_attributes, length, segment_length, limits);
}

template <unsigned IndexSize, typename Uncompressed, typename Compressed, typename UnaryModelFunction>
void kernel_t<IndexSize, Uncompressed, Compressed, UnaryModelFunction>::enqueue_launch_with(
	stream::id_t                   stream,
	const launch_configuration_t&  launch_config,
	const arguments_type&                 arguments) const
{
	auto decompressed               = argument<Uncompressed*>(arguments, "decompressed");
	auto compressed_input           = argument<const Compressed*>(arguments, "compressed_input");
	auto segment_model_coefficients = argument<const model_coefficients_type*>(arguments, "segment_model_coefficients");
	auto patch_values               = argument<const Uncompressed*>(arguments, "patch_values");
	auto length                     = argument<index_type>(arguments, "length");
	auto segment_length             = argument<index_type>(arguments, "segment_length");
	auto num_patches                = argument<index_type>(arguments, "num_patches");
	auto patch_positions = decompression::patched::positions<IndexSize>(stream, arguments, num_patches);   // plain, or expanded from compressed ones

	// (where the plain decoder followed by the scatter is the faster way — 2-byte outputs, bands of patch densities: patched/decode_then_scatter.hpp)
	if constexpr (sizeof(Uncompressed) == 2) {
		if (decompression::patched::decode_then_scatter_is_faster(decompression::patched::family_t::frame_of_reference, 2, (uint64_t) num_patches, (uint64_t) length)) {
			arguments_type resolution;
			resolution["length"] = (size_t) length;
			resolution["segment_length"] = (size_t) segment_length;
			decompression::patched::decode_then_scatter<kernel_t<IndexSize, Uncompressed, Compressed, UnaryModelFunction>, IndexSize, 2>(
				stream, launch_config, arguments, resolution, reinterpret_cast<uint_t<2>*>(decompressed), reinterpret_cast<const uint_t<2>*>(patch_values), patch_positions, (size_type) num_patches);
			return;
		}
	}

	// (columns with several patches to a row of 64 lanes' elements: the build in which every lane looks for its own patches — on_device/sorted_patches.hpp)
	const bool many_patches = sorted_patches::many_patches_per_row((uint64_t) num_patches, (uint64_t) length, 16 / (unsigned) sizeof(Uncompressed), sorted_patches::together_from::frame_of_reference);
	cuda::enqueue_launch(many_patches ? reinterpret_cast<const void*>(patched::decompress<IndexSize, Uncompressed, Compressed, UnaryModelFunction, true>) : reinterpret_cast<const void*>(patched::decompress<IndexSize, Uncompressed, Compressed, UnaryModelFunction>), stream, launch_config,
		decompressed, compressed_input, segment_model_coefficients, patch_values, patch_positions,
		(size_type) length, (size_type) segment_length, (size_type) num_patches);
}

template <unsigned IndexSize, typename Uncompressed, typename Compressed, typename UnaryModelFunction>
const device_function_t kernel_t<IndexSize, Uncompressed, Compressed, UnaryModelFunction>::get_device_function() const
{
	return device_function_t(reinterpret_cast<const void*>(patched::decompress<IndexSize, Uncompressed, Compressed, UnaryModelFunction>), { reinterpret_cast<const void*>(patched::decompress<IndexSize, Uncompressed, Compressed, UnaryModelFunction, true>) });
}

} // namespace patched
} // namespace frame_of_reference
} // namespace decompression
} // namespace kernels
} // namespace cuda
