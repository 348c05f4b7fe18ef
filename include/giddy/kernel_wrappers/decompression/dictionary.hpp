// Kernel wrapper for Dictionary decompression.  Argument-map keys and exact types follow the
// reference's src/kernel_wrappers/decompression/dictionary.cu:52-96, including its two different
// spellings of the caching flag (resolve: "cache_input_data_in_shared_mem", launch:
// "cache_dictionary_in_shared_mem"; SURVEY.md quirks register).  Unlike the reference, leaving the
// flag out stages the dictionary in LDS whenever it fits.
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/decompression/dictionary.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace dictionary {

template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
	REGISTERED_KERNEL_WRAPPER_TYPED_LAUNCH(kernel_t);
	REGISTERED_KERNEL_WRAPPER_BATCHES(kernel_t);

	using dictionary_index_type = uint_t<DictionaryIndexSize>;
	using dictionary_size_type  = size_type_by_index_size<DictionaryIndexSize>;
	using uncompressed_type     = uint_t<UncompressedSize>;
	using size_type             = size_type_by_index_size<IndexSize>;

	/// THE decode on `stream` — what the type-erased enqueue_launch(), cuda::kernel::enqueue_launch(kernel, typed parameters...) and dictionary::launch<...>()
	/// all run: `decompress`, or its build for columns at any element offset when the addresses ask for it (kernels/decompression/dictionary.hpp)
	static void launch(stream::id_t stream, const launch_configuration_t& launch_config,
		uncompressed_type* decompressed, const dictionary_index_type* compressed_input, const uncompressed_type* dictionary_entries,
		size_type length, dictionary_size_type num_dictionary_entries, bool cache_dictionary_in_shared_memory, serialization_factor_t serialization_factor)
	{
		const void* kernel = stream_tiles::wants_any_alignment_build(decompressed, compressed_input)
			? reinterpret_cast<const void*>(decompress_at_any_alignment<IndexSize, UncompressedSize, DictionaryIndexSize>)
			: reinterpret_cast<const void*>(decompress<IndexSize, UncompressedSize, DictionaryIndexSize>);
		cuda::enqueue_launch(kernel, stream, launch_config,
			decompressed, compressed_input, dictionary_entries, length, num_dictionary_entries, cache_dictionary_in_shared_memory, serialization_factor);
	}

	/// the argument map unpacked: sink(launch()'s parameters) — for the one launch and for a piece of a batch alike
	template <typename Sink>
	static void with_typed_parameters(const launch_configuration_t& launch_config, const arguments_type& arguments, Sink&& sink);

	/// a column as one piece of a batch (giddy/batch.hpp); hints[3]: some piece lies where `decompress` would go element by element
	static void add_piece(batch::table_t& table, const launch_configuration_t& piece_config,
		uncompressed_type* decompressed, const dictionary_index_type* compressed_input, const uncompressed_type* dictionary_entries,
		size_type length, dictionary_size_type num_dictionary_entries, bool cache_dictionary_in_shared_memory, serialization_factor_t serialization_factor)
	{
		if (stream_tiles::wants_any_alignment_build(decompressed, compressed_input)) { table.hints[3] = 1; }
		table.add(piece_config, decompressed, compressed_input, dictionary_entries, length, num_dictionary_entries, cache_dictionary_in_shared_memory, serialization_factor);
	}

	/// ... and the batch's ONE launch
	static void launch_batch(stream::id_t stream, const batch::on_device_t& tables)
	{
		const void* kernel = tables.hints[3] != 0
			? reinterpret_cast<const void*>(batch::decode_pieces<piece_body<IndexSize, UncompressedSize, DictionaryIndexSize, true>>)
			: reinterpret_cast<const void*>(batch::decode_pieces<piece_body<IndexSize, UncompressedSize, DictionaryIndexSize, false>>);
		cuda::enqueue_launch(kernel, stream, launch_configuration_t(tables.num_blocks, tables.block_size, tables.dynamic_shared_memory_size), tables.pieces, tables.blocks);
	}

	launch_configuration_t resolve_launch_configuration(
		device::properties_t              device_properties,
		device_function::attributes_t     kernel_function_attributes,
		size_t                            num_dictionary_entries,
		size_t                            data_length,
		optional<bool>                    cache_input_data_in_shared_mem,
		optional<serialization_factor_t>  serialization_factor,
		launch_configuration_limits_t     limits) const
	{
		launch_config_resolution_params_t<IndexSize, UncompressedSize, DictionaryIndexSize> params(
			device_properties, kernel_function_attributes, num_dictionary_entries, data_length,
			cache_input_data_in_shared_mem, limits.dynamic_shared_memory);
		return cuda::kernels::resolve_launch_configuration(params, limits, serialization_factor);
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
	auto serialization_factor   = optional_argument<serialization_factor_t>(extra_arguments, "serialization_factor");
	return resolve_launch_configuration(device_properties, kernel_function_attributes,
		num_dictionary_entries, data_length, cache_input_data_in_shared_mem, serialization_factor, limits);
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize>
void kernel_t<IndexSize, UncompressedSize, DictionaryIndexSize>::enqueue_launch_with(
	stream::id_t                   stream,
	const launch_configuration_t&  launch_config,
	const arguments_type&                 arguments) const
{
	with_typed_parameters(launch_config, arguments, [&](auto... parameters) { launch(stream, launch_config, parameters...); });
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize>
template <typename Sink>
void kernel_t<IndexSize, UncompressedSize, DictionaryIndexSize>::with_typed_parameters(
	const launch_configuration_t&  launch_config,
	const arguments_type&          arguments,
	Sink&&                         sink)
{
	auto maybe_cache = optional_argument<bool>(arguments, "cache_dictionary_in_shared_mem");
	auto maybe_serialization_factor = optional_argument<serialization_factor_t>(arguments, "serialization_factor");

	auto decompressed            = argument<uncompressed_type*>(arguments, "decompressed");
	auto compressed_input        = argument<const dictionary_index_type*>(arguments, "compressed_input");
	auto dictionary_entries      = argument<const uncompressed_type*>(arguments, "dictionary_entries");
	auto length                  = argument<size_type>(arguments, "length");
	auto num_dictionary_entries  = argument<dictionary_size_type>(arguments, "num_dictionary_entries");

	// The dictionary is staged in LDS iff the resolved configuration reserved room for it.
	bool fits = (size_t) num_dictionary_entries * UncompressedSize <= launch_config.dynamic_shared_memory_size;
	bool caching = maybe_cache.value_or(launch_config.dynamic_shared_memory_size > 0) && fits;
	auto serialization_factor = maybe_serialization_factor.value_or(get_default_serialization_factor(caching));

	sink(decompressed, compressed_input, dictionary_entries, length, num_dictionary_entries, caching, serialization_factor);
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize>
const device_function_t kernel_t<IndexSize, UncompressedSize, DictionaryIndexSize>::get_device_function() const
{
	// (the any-alignment build takes the configuration resolved for `decompress` — same LDS, no launch bounds; kept out of the merged attributes so that
	// the configuration of the aligned column is what it has been)
	return device_function_t(reinterpret_cast<const void*>(decompress<IndexSize, UncompressedSize, DictionaryIndexSize>), { },
		{ reinterpret_cast<const void*>(decompress_at_any_alignment<IndexSize, UncompressedSize, DictionaryIndexSize>) });
}

/// kernel_t<...>::launch as a free function: the one-line replacement for a direct launch of `decompress` ("mode C")
template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize>
inline void launch(stream::id_t stream, const launch_configuration_t& launch_config,
	uint_t<UncompressedSize>* decompressed, const uint_t<DictionaryIndexSize>* compressed_input, const uint_t<UncompressedSize>* dictionary_entries,
	size_type_by_index_size<IndexSize> length, size_type_by_index_size<DictionaryIndexSize> num_dictionary_entries,
	bool cache_dictionary_in_shared_memory, serialization_factor_t serialization_factor)
{
	kernel_t<IndexSize, UncompressedSize, DictionaryIndexSize>::launch(stream, launch_config, decompressed, compressed_input, dictionary_entries,
		length, num_dictionary_entries, cache_dictionary_in_shared_memory, serialization_factor);
}

} // namespace dictionary
} // namespace decompression
} // namespace kernels
} // namespace cuda
