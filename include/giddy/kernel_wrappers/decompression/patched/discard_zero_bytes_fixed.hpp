// Kernel wrapper for fixed-width discard-zero-bytes decompression fused with patching (an extension, SURVEY.md §8f.1;
// the reference runs discard_zero_bytes_fixed.cu and then data_layout/scatter.cu).  Argument-map keys: those of the
// DZB-fixed wrapper (src/kernel_wrappers/decompression/discard_zero_bytes_fixed.cu:60-97) plus "patch_values",
// "patch_positions" (ascending) and "num_patches" — or, in place of "patch_positions", the compressed-indices arrays and
// a "patch_positions_scratch" (patched/patch_positions.hpp).
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernel_wrappers/decompression/patched/patch_positions.hpp"
#include "giddy/kernels/decompression/discard_zero_bytes_fixed.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace discard_zero_bytes {
namespace fixed_width {
namespace patched {

using util::endianness_t;

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize,
	endianness_t UncompressedEndianness = util::compilation_target_endianness(),
	serialization_factor_t SerializationFactor = DefaultSerializationFactor>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);

	using uncompressed_type = uint_t<UncompressedSize>;
	using compressed_type   = uint_t<CompressedSize>;
	using index_type        = uint_t<IndexSize>;
	using size_type         = size_type_by_index_size<IndexSize>;

	launch_configuration_t resolve_launch_configuration(
		device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
		size_t length, launch_configuration_limits_t limits, size_t num_patches = 0) const
	{
		fixed_width::launch_config_resolution_params_t<IndexSize, UncompressedSize, CompressedSize, translate(UncompressedEndianness), SerializationFactor>
			params(device_properties, kernel_function_attributes, length);
		// a thread of the patched kernel covers a wavefront chunk's share of elements rather than a tile's
		params.length = util::div_rounding_up(length, (size_t) stream_tiles::elements_per_thread_per_chunk<uncompressed_type>());
		// resident workgroups per CU, measured per instantiation (tools/occupancy_sweep.py): 8-byte outputs 2 (72 -> 77-81 % of HBM
		// peak; 1- and 2-byte inputs are staged through LDS), <4,2> and <2,1> 4 (71-74 -> 77), <4,1> unthrottled (80)
		params.resident_blocks_per_multiprocessor_cap = nullopt;
		if (UncompressedSize == 8) { params.resident_blocks_per_multiprocessor_cap = 2u; }
		else if (!(UncompressedSize == 4 && CompressedSize == 1)) { params.resident_blocks_per_multiprocessor_cap = 4u; }
		// (... for columns with few patches.  One whose rows wait for windows of patches from memory wants every wavefront it can have: a caller that states "num_patches"
		// when resolving is not throttled where the ManyPatches build will be launched — <4,2,1> at 2 / 5 / 20 % of patches: 51 -> 69 / 45 -> 63 / 33 -> 47 % of the HBM peak,
		// <4,8,1> at 5 / 20 %: 45 -> 70 / 41 -> 67, <4,4,2>: 66 -> 77 / 57 -> 70 — docs/history/r05.md)
		if (sorted_patches::many_patches_per_row((uint64_t) num_patches, (uint64_t) length, 16 / UncompressedSize, sorted_patches::together_from::discard_zero_bytes_fixed)) {
			params.resident_blocks_per_multiprocessor_cap = nullopt;
		}
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}
};

#define GIDDY_DZBFP_TEMPLATE template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize, endianness_t UncompressedEndianness, serialization_factor_t SerializationFactor>
#define GIDDY_DZBFP_KERNEL kernel_t<IndexSize, UncompressedSize, CompressedSize, UncompressedEndianness, SerializationFactor>

GIDDY_DZBFP_TEMPLATE
launch_configuration_t GIDDY_DZBFP_KERNEL::resolve_launch_configuration(
	device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
	arguments_type extra_arguments, launch_configuration_limits_t limits) const
{
	auto length = argument<size_t>(extra_arguments, "length");
	auto num_patches = optional_argument<size_t>(extra_arguments, "num_patches");   // (optional: picks the residency)
	return resolve_launch_configuration(device_properties, kernel_function_attributes, length, limits, num_patches.value_or(0));
}

GIDDY_DZBFP_TEMPLATE
void GIDDY_DZBFP_KERNEL::enqueue_launch_with(stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	auto decompressed     = argument<uncompressed_type*>(arguments, "decompressed");
	auto compressed_input = argument<const compressed_type*>(arguments, "compressed_input");
	auto patch_values     = argument<const uncompressed_type*>(arguments, "patch_values");
	auto length           = argument<index_type>(arguments, "length");
	auto num_patches      = argument<index_type>(arguments, "num_patches");
	auto patch_positions = decompression::patched::positions<IndexSize>(stream, arguments, num_patches);   // plain, or expanded from compressed ones
	// (columns with several patches to a row of 64 lanes' elements: the build in which every lane looks for its own patches — on_device/sorted_patches.hpp)
	const bool many_patches = sorted_patches::many_patches_per_row((uint64_t) num_patches, (uint64_t) length, 16 / UncompressedSize, sorted_patches::together_from::discard_zero_bytes_fixed);
	cuda::enqueue_launch(many_patches ? reinterpret_cast<const void*>(patched::decompress<IndexSize, UncompressedSize, CompressedSize, translate(UncompressedEndianness), SerializationFactor, true>)
		: reinterpret_cast<const void*>(patched::decompress<IndexSize, UncompressedSize, CompressedSize, translate(UncompressedEndianness), SerializationFactor>), stream, launch_config,
		decompressed, compressed_input, patch_values, patch_positions, (size_type) length, (size_type) num_patches);
}

GIDDY_DZBFP_TEMPLATE
const device_function_t GIDDY_DZBFP_KERNEL::get_device_function() const
{
	return device_function_t(reinterpret_cast<const void*>(patched::decompress<IndexSize, UncompressedSize, CompressedSize, translate(UncompressedEndianness), SerializationFactor>),
		{ reinterpret_cast<const void*>(patched::decompress<IndexSize, UncompressedSize, CompressedSize, translate(UncompressedEndianness), SerializationFactor, true>) });
}

#undef GIDDY_DZBFP_TEMPLATE
#undef GIDDY_DZBFP_KERNEL

} // namespace patched
} // namespace fixed_width
} // namespace discard_zero_bytes
} // namespace decompression
} // namespace kernels
} // namespace cuda
