// Kernel wrapper for Model decompression fused with patching (an extension; the reference runs
// model.cu and then data_layout/scatter.cu).  Argument-map keys: those of the Model wrapper
// (src/kernel_wrappers/decompression/model.cu:44-75) plus "patch_values", "patch_positions" (ascending) and
// "num_patches" — or, in place of "patch_positions", the compressed-indices arrays and a "patch_positions_scratch"
// (patched/patch_positions.hpp).
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernel_wrappers/decompression/patched/patch_positions.hpp"
#include "giddy/kernel_wrappers/decompression/patched/decode_then_scatter.hpp"
#include "giddy/kernels/decompression/model.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace model {
namespace patched {

template <unsigned IndexSize, typename Uncompressed, typename UnaryModelFunction>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);

	using index_type = uint_t<IndexSize>;
	using size_type  = size_type_by_index_size<IndexSize>;

	launch_configuration_t resolve_launch_configuration(
		device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
		size_t length, launch_configuration_limits_t limits, size_t num_patches = 0) const
	{
		model::launch_config_resolution_params_t<IndexSize, Uncompressed, UnaryModelFunction> params(
			device_properties, kernel_function_attributes, length);
		// a thread of the patched kernel covers a wavefront chunk's share of elements rather than a tile's
		params.length = util::div_rounding_up(length, (size_t) stream_tiles::group_size<Uncompressed>::value * GIDDY_MODEL_PATCHED_ROWS_PER_CHUNK);
		// (a store stream with few patches likes few wavefronts; one whose every row waits for a window of patches from memory wants them all: a caller that
		// states "num_patches" when resolving gets the residency for its density — from 3 patches to a row of 64 lanes' elements on: 5 workgroups per CU, from 16: no throttle)
		const uint64_t patches_x_row = (uint64_t) num_patches * (64 * stream_tiles::group_size<Uncompressed>::value);
		const bool many = patches_x_row >= (stream_tiles::group_size<Uncompressed>::value >= 8 ? 2 : 3) * (uint64_t) length, dense   /* (2-byte outputs: wherever their ManyPatches build is taken - at 1 % of patches 3 | 5 workgroups: 41-44 | 58-62 % of the HBM peak) */ = patches_x_row >= 16 * (uint64_t) length;
		params.resident_blocks_per_multiprocessor_cap = (GIDDY_MODEL_PATCHED_RESIDENT_BLOCKS_CAP > 0 && !dense)
			? optional<unsigned>(many ? 5u : (unsigned) GIDDY_MODEL_PATCHED_RESIDENT_BLOCKS_CAP) : optional<unsigned>(nullopt);
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}
};

template <unsigned IndexSize, typename Uncompressed, typename UnaryModelFunction>
launch_configuration_t kernel_t<IndexSize, Uncompressed, UnaryModelFunction>::resolve_launch_configuration(
	device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
	arguments_type extra_arguments, launch_configuration_limits_t limits) const
{
	auto length = argument<size_t>(extra_arguments, "length");
	auto num_patches = optional_argument<size_t>(extra_arguments, "num_patches");   // (optional: see above)
	return resolve_launch_configuration(device_properties, kernel_function_attributes, length, limits, num_patches.value_or(0));
}

template <unsigned IndexSize, typename Uncompressed, typename UnaryModelFunction>
void kernel_t<IndexSize, Uncompressed, UnaryModelFunction>::enqueue_launch_with(
	stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	using model_coefficients_type = typename UnaryModelFunction::coefficients_type;
	auto length             = argument<index_type>(arguments, "length");
	auto decompressed       = argument<Uncompressed*>(arguments, "decompressed");
	auto model_coefficients = argument<model_coefficients_type>(arguments, "model_coefficients");
	auto patch_values       = argument<const Uncompressed*>(arguments, "patch_values");
	auto num_patches        = argument<index_type>(arguments, "num_patches");
	auto patch_positions = decompression::patched::positions<IndexSize>(stream, arguments, num_patches);   // plain, or expanded from compressed ones

	// (where the plain decoder followed by the scatter is the faster way — 2-byte outputs, bands of patch densities: patched/decode_then_scatter.hpp)
	if constexpr (sizeof(Uncompressed) == 2) {
		if (decompression::patched::decode_then_scatter_is_faster(decompression::patched::family_t::model, 2, (uint64_t) num_patches, (uint64_t) length)) {
			arguments_type resolution;
			resolution["length"] = (size_t) length;
			decompression::patched::decode_then_scatter<kernel_t<IndexSize, Uncompressed, UnaryModelFunction>, IndexSize, 2>(
				stream, launch_config, arguments, resolution, reinterpret_cast<uint_t<2>*>(decompressed), reinterpret_cast<const uint_t<2>*>(patch_values), patch_positions, (size_type) num_patches);
			return;
		}
	}
	// (columns with several patches to a row of 64 lanes' elements: the build in which every lane looks for its own patches — on_device/sorted_patches.hpp)
	const bool many_patches = sorted_patches::many_patches_per_row((uint64_t) num_patches, (uint64_t) length, 16 / (unsigned) sizeof(Uncompressed), sorted_patches::together_from::model);
	cuda::enqueue_launch(many_patches ? reinterpret_cast<const void*>(patched::decompress<IndexSize, Uncompressed, UnaryModelFunction, true>) : reinterpret_cast<const void*>(patched::decompress<IndexSize, Uncompressed, UnaryModelFunction>), stream, launch_config,
		decompressed, model_coefficients, patch_values, patch_positions, (size_type) length, (size_type) num_patches);
}

template <unsigned IndexSize, typename Uncompressed, typename UnaryModelFunction>
const device_function_t kernel_t<IndexSize, Uncompressed, UnaryModelFunction>::get_device_function() const
{
	return device_function_t(reinterpret_cast<const void*>(patched::decompress<IndexSize, Uncompressed, UnaryModelFunction>), { reinterpret_cast<const void*>(patched::decompress<IndexSize, Uncompressed, UnaryModelFunction, true>) });
}

} // namespace patched
} // namespace model
} // namespace decompression
} // namespace kernels
} // namespace cuda
