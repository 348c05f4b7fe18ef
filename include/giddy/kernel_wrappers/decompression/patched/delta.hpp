// Kernel wrapper for Delta decompression fused with patching (an extension; the
// reference runs delta.cu and then data_layout/scatter.cu).  Argument-map keys: those of the Delta wrapper
// (src/kernel_wrappers/decompression/delta.cu:49-83) plus the patch arrays under the names "patch_values",
// "patch_positions" (ascending) and "num_patches" — or, in place of "patch_positions", the compressed-indices arrays
// and a "patch_positions_scratch" (patched/patch_positions.hpp).
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernel_wrappers/decompression/patched/patch_positions.hpp"
#include "giddy/kernel_wrappers/decompression/patched/decode_then_scatter.hpp"
#include "giddy/kernels/decompression/delta.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace delta {
namespace patched {

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);

	using compressed_type   = uint_t<CompressedSize>;
	using uncompressed_type = uint_t<UncompressedSize>;
	using index_type        = uint_t<IndexSize>;
	using size_type         = size_type_by_index_size<IndexSize>;

	/// the fused kernel is launched in the shape of the plain one: the patches ride along
	launch_configuration_t resolve_launch_configuration(
		device::properties_t           device_properties,
		device_function::attributes_t  kernel_function_attributes,
		size_t                         length,
		size_t                         segment_length,
		launch_configuration_limits_t  limits,
		size_t                         num_patches = 0) const
	{
		delta::launch_config_resolution_params_t<IndexSize, UncompressedSize, CompressedSize> params(
			device_properties, kernel_function_attributes, length, segment_length, limits.dynamic_shared_memory);
		// (the plain kernel's throttle — 1-byte deltas of 8-byte elements: 4 resident workgroups per CU — suits columns with few patches; where the ManyPatches build will be
		// launched, known when the caller states "num_patches" when resolving, the rows wait for windows of patches from memory and want every wavefront: as the other fused wrappers)
		if (sorted_patches::many_patches_per_row((uint64_t) num_patches, (uint64_t) length, 16 / UncompressedSize, sorted_patches::together_from::delta)) {
			params.resident_blocks_per_multiprocessor_cap = nullopt;
		}
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
	auto num_patches    = optional_argument<size_t>(extra_arguments, "num_patches");   // (optional: picks the residency)
	return resolve_launch_configuration(device_properties, kernel_function_attributes, length, segment_length, limits, num_patches.value_or(0));
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
void kernel_t<IndexSize, UncompressedSize, CompressedSize>::enqueue_launch_with(
	stream::id_t                   stream,
	const launch_configuration_t&  launch_config,
	const arguments_type&                 arguments) const
{
	auto decompressed      = argument<uncompressed_type*>(arguments, "decompressed");
	auto compressed_input  = argument<const compressed_type*>(arguments, "compressed_input");
	auto anchor_values     = argument<const uncompressed_type*>(arguments, "anchor_values");
	auto patch_values      = argument<const uncompressed_type*>(arguments, "patch_values");
	auto length            = argument<index_type>(arguments, "length");
	auto segment_length    = argument<index_type>(arguments, "segment_length");
	auto num_patches       = argument<index_type>(arguments, "num_patches");
	auto patch_positions = decompression::patched::positions<IndexSize>(stream, arguments, num_patches);   // plain, or expanded from compressed ones

	// (where the plain decoder followed by the scatter is the faster way — 2-byte outputs, bands of patch densities: patched/decode_then_scatter.hpp)
	if constexpr (UncompressedSize == 2) {
		if (decompression::patched::decode_then_scatter_is_faster(decompression::patched::family_t::delta, UncompressedSize, (uint64_t) num_patches, (uint64_t) length)) {
			arguments_type resolution;
			resolution["length"] = (size_t) length;
			resolution["segment_length"] = (size_t) segment_length;
			decompression::patched::decode_then_scatter<kernel_t<IndexSize, UncompressedSize, CompressedSize>, IndexSize, UncompressedSize>(
				stream, launch_config, arguments, resolution, decompressed, patch_values, patch_positions, (size_type) num_patches);
			return;
		}
	}

	// (columns with several patches to a row of 64 lanes' elements: the build in which every lane looks for its own patches — on_device/sorted_patches.hpp)
	const bool many_patches = sorted_patches::many_patches_per_row((uint64_t) num_patches, (uint64_t) length, 16 / UncompressedSize, sorted_patches::together_from::delta);
	cuda::enqueue_launch(many_patches ? reinterpret_cast<const void*>(patched::decompress<IndexSize, UncompressedSize, CompressedSize, true>) : reinterpret_cast<const void*>(patched::decompress<IndexSize, UncompressedSize, CompressedSize>), stream, launch_config,
		decompressed, compressed_input, anchor_values, patch_values, patch_positions,
		(size_type) length, (size_type) segment_length, (size_type) num_patches);
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
const device_function_t kernel_t<IndexSize, UncompressedSize, CompressedSize>::get_device_function() const
{
	return device_function_t(reinterpret_cast<const void*>(patched::decompress<IndexSize, UncompressedSize, CompressedSize>), { reinterpret_cast<const void*>(patched::decompress<IndexSize, UncompressedSize, CompressedSize, true>) });
}

} // namespace patched
} // namespace delta
} // namespace decompression
} // namespace kernels
} // namespace cuda
