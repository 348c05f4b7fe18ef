// Kernel wrapper for Dictionary decompression fused with patching (an extension).
// Argument-map keys: those of the Dictionary wrapper (src/kernel_wrappers/decompression/dictionary.cu:52-96,
// both spellings of the caching flag included) plus "patch_values", "patch_positions" (ascending) and
// "num_patches" — or, in place of "patch_positions", the compressed-indices arrays and a "patch_positions_scratch"
// (patched/patch_positions.hpp).
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernel_wrappers/decompression/patched/patch_positions.hpp"
#include "giddy/kernel_wrappers/decompression/patched/decode_then_scatter.hpp"
#include "giddy/kernels/decompression/dictionary.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace dictionary {
namespace patched {

template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);

	using dictionary_index_type = uint_t<DictionaryIndexSize>;
	using dictionary_size_type  = size_type_by_index_size<DictionaryIndexSize>;
	using uncompressed_type     = uint_t<UncompressedSize>;
	using index_type            = uint_t<IndexSize>;
	using size_type             = size_type_by_index_size<IndexSize>;

	launch_configuration_t resolve_launch_configuration(
		device::properties_t              device_properties,
		device_function::attributes_t     kernel_function_attributes,
		size_t                            num_dictionary_entries,
		size_t                            data_length,
		optional<bool>                    cache_input_data_in_shared_mem,
		launch_configuration_limits_t     limits,
		size_t                            num_patches = 0) const
	{
		dictionary::launch_config_resolution_params_t<IndexSize, UncompressedSize, DictionaryIndexSize> params(
			device_properties, kernel_function_attributes, num_dictionary_entries, data_length,
			cache_input_data_in_shared_mem, limits.dynamic_shared_memory);
		// a thread of the patched kernel covers a wavefront chunk's share of elements rather than a tile's
		params.length = util::div_rounding_up(data_length, (size_t) stream_tiles::elements_per_thread_per_chunk<uncompressed_type>());
		params.resident_blocks_per_multiprocessor_cap = nullopt;   // (the plain kernel's occupancy throttle was not measured for this one)
		params.reserve_chunk_key_staging();   // (narrow keys of 8-byte values: staged through LDS, 2 resident workgroups per CU)
		// (... for columns with few patches: where the ManyPatches build will be launched — known when the caller states "num_patches" when resolving — the rows wait for
		// windows of patches from memory and want every wavefront they can have; as fused fixed-width discard-zero-bytes, patched/discard_zero_bytes_fixed.hpp)
		if (sorted_patches::many_patches_per_row((uint64_t) num_patches, (uint64_t) data_length, 16 / UncompressedSize, sorted_patches::together_from::dictionary)) {
			params.resident_blocks_per_multiprocessor_cap = nullopt;
		}
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}
};

template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize>
launch_configuration_t kernel_t<IndexSize, UncompressedSize, DictionaryIndexSize>::resolve_launch_configuration(
	device::properties_t           device_properties,
	device_function::attributes_t  kernel_function_attributes,
	arguments_type                 extra_arguments,
	launch_configuration_limits_t  limits) const
{
	auto num_dictionary_entries = argument<size_t>(extra_arguments, "num_dictionary_entries");
	auto data_length            = argument<size_t>(extra_arguments, "data_length");
	auto cache_input_data_in_shared_mem = optional_argument<bool>(extra_arguments, "cache_input_data_in_shared_mem");
	auto num_patches = optional_argument<size_t>(extra_arguments, "num_patches");   // (optional: picks the residency)
	return resolve_launch_configuration(device_properties, kernel_function_attributes,
		num_dictionary_entries, data_length, cache_input_data_in_shared_mem, limits, num_patches.value_or(0));
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize>
void kernel_t<IndexSize, UncompressedSize, DictionaryIndexSize>::enqueue_launch_with(
	stream::id_t                   stream,
	const launch_configuration_t&  launch_config,
	const arguments_type&                 arguments) const
{
	auto maybe_cache = optional_argument<bool>(arguments, "cache_dictionary_in_shared_mem");

	auto decompressed            = argument<uncompressed_type*>(arguments, "decompressed");
	auto compressed_input        = argument<const dictionary_index_type*>(arguments, "compressed_input");
	auto dictionary_entries      = argument<const uncompressed_type*>(arguments, "dictionary_entries");
	auto patch_values            = argument<const uncompressed_type*>(arguments, "patch_values");
	auto length                  = argument<size_type>(arguments, "length");
	auto num_dictionary_entries  = argument<dictionary_size_type>(arguments, "num_dictionary_entries");
	auto num_patches             = argument<index_type>(arguments, "num_patches");
	auto patch_positions = decompression::patched::positions<IndexSize>(stream, arguments, num_patches);   // plain, or expanded from compressed ones

	// The dictionary is staged in LDS iff the resolved configuration reserved room for it.
	bool fits = (size_t) num_dictionary_entries * UncompressedSize <= launch_config.dynamic_shared_memory_size;
	bool caching = maybe_cache.value_or(launch_config.dynamic_shared_memory_size > 0) && fits;

	// (2-byte keys of 8-byte values: a chunk's keys go through dynamic LDS behind the dictionary iff the configuration has the room)
	size_t behind = caching ? ((size_t) num_dictionary_entries * UncompressedSize + 15) / 16 * 16 : 0;
	bool chunk_key_staging = DictionaryIndexSize > 1 && dictionary::detail::stages_chunk_keys<UncompressedSize, DictionaryIndexSize>() &&
		launch_config.dynamic_shared_memory_size >= behind + sizeof(dictionary::detail::chunk_key_storage<UncompressedSize, DictionaryIndexSize>);

	// (columns with several patches to a row of 64 lanes' elements: the build in which every lane looks for its own patches — on_device/sorted_patches.hpp)
	const bool many_patches = sorted_patches::many_patches_per_row((uint64_t) num_patches, (uint64_t) length, 16 / UncompressedSize, sorted_patches::together_from::dictionary);
	// (where the plain decoder followed by the scatter is the faster way — 8-byte values beside a staged dictionary of 16 KiB or more in the ManyPatches build; 2-byte values
	// from a third of the elements patched on: patched/decode_then_scatter.hpp)
	if constexpr (UncompressedSize == 8 || UncompressedSize == 2) {
		if (decompression::patched::decode_then_scatter_is_faster(decompression::patched::family_t::dictionary, UncompressedSize, (uint64_t) num_patches, (uint64_t) length,
			caching ? (size_t) num_dictionary_entries * UncompressedSize : 0, many_patches)) {
			arguments_type resolution;
			resolution["num_dictionary_entries"] = (size_t) num_dictionary_entries;
			resolution["data_length"] = (size_t) length;
			if (maybe_cache) { resolution["cache_input_data_in_shared_mem"] = maybe_cache.value(); }
			decompression::patched::decode_then_scatter<kernel_t<IndexSize, UncompressedSize, DictionaryIndexSize>, IndexSize, UncompressedSize>(
				stream, launch_config, arguments, resolution, decompressed, patch_values, patch_positions, (size_type) num_patches);
			return;
		}
	}
	cuda::enqueue_launch(many_patches ? reinterpret_cast<const void*>(patched::decompress<IndexSize, UncompressedSize, DictionaryIndexSize, true>) : reinterpret_cast<const void*>(patched::decompress<IndexSize, UncompressedSize, DictionaryIndexSize>), stream, launch_config,
		decompressed, compressed_input, dictionary_entries, patch_values, patch_positions,
		length, num_dictionary_entries, (size_type) num_patches, caching, chunk_key_staging);
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize>
const device_function_t kernel_t<IndexSize, UncompressedSize, DictionaryIndexSize>::get_device_function() const
{
	return device_function_t(reinterpret_cast<const void*>(patched::decompress<IndexSize, UncompressedSize, DictionaryIndexSize>), { reinterpret_cast<const void*>(patched::decompress<IndexSize, UncompressedSize, DictionaryIndexSize, true>) });
}

} // namespace patched
} // namespace dictionary
} // namespace decompression
} // namespace kernels
} // namespace cuda
