// Kernel wrapper for Frame-of-Reference decompression: typed and type-erased launch-configuration
// resolution, type-erased enqueue, factory registration.  Argument-map keys and their exact C++
// types follow the reference's src/kernel_wrappers/decompression/frame_of_reference.cu:58-102.
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/decompression/frame_of_reference.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace frame_of_reference {

template <unsigned IndexSize, typename Uncompressed, typename Compressed, typename UnaryModelFunction>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
	REGISTERED_KERNEL_WRAPPER_TYPED_LAUNCH(kernel_t);
	REGISTERED_KERNEL_WRAPPER_BATCHES(kernel_t);

	using size_type               = size_type_by_index_size<IndexSize>;
	using model_coefficients_type = typename UnaryModelFunction::coefficients_type;

	/// THE decode on `stream` — what the type-erased enqueue_launch(), cuda::kernel::enqueue_launch(kernel, typed parameters...) and frame_of_reference::launch<...>()
	/// all run: `decompress`, or its build for columns at any element offset when the addresses ask for it (kernels/decompression/frame_of_reference.hpp)
	static void launch(stream::id_t stream, const launch_configuration_t& launch_config,
		Uncompressed* decompressed, const Compressed* compressed_input, const model_coefficients_type* segment_model_coefficients,
		size_type length, size_type segment_length, uint_t<IndexSize> num_segments_to_decompress_per_block)
	{
		const void* kernel = stream_tiles::wants_any_alignment_build(decompressed, compressed_input)
			? reinterpret_cast<const void*>(decompress_at_any_alignment<IndexSize, Uncompressed, Compressed, UnaryModelFunction>)
			: reinterpret_cast<const void*>(decompress<IndexSize, Uncompressed, Compressed, UnaryModelFunction>);
		cuda::enqueue_launch(kernel, stream, launch_config,
			decompressed, compressed_input, segment_model_coefficients, length, segment_length, num_segments_to_decompress_per_block);
	}

	/// the argument map unpacked: sink(launch()'s parameters) — for the one launch and for a piece of a batch alike
	template <typename Sink>
	static void with_typed_parameters(const launch_configuration_t& launch_config, const arguments_type& arguments, Sink&& sink);

	/// a column as one piece of a batch (giddy/batch.hpp); hints[3]: some piece lies where `decompress` would go element by element
	static void add_piece(batch::table_t& table, const launch_configuration_t& piece_config,
		Uncompressed* decompressed, const Compressed* compressed_input, const model_coefficients_type* segment_model_coefficients,
		size_type length, size_type segment_length, uint_t<IndexSize> num_segments_to_decompress_per_block)
	{
		if (segment_length == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
		if (stream_tiles::wants_any_alignment_build(decompressed, compressed_input)) { table.hints[3] = 1; }
		table.add(piece_config, decompressed, compressed_input, segment_model_coefficients, length, segment_length, num_segments_to_decompress_per_block);
	}

	/// ... and the batch's ONE launch
	static void launch_batch(stream::id_t stream, const batch::on_device_t& tables)
	{
		const void* kernel = tables.hints[3] != 0
			? reinterpret_cast<const void*>(batch::decode_pieces<piece_body<IndexSize, Uncompressed, Compressed, UnaryModelFunction, true>>)
			: reinterpret_cast<const void*>(batch::decode_pieces<piece_body<IndexSize, Uncompressed, Compressed, UnaryModelFunction, false>>);
		cuda::enqueue_launch(kernel, stream, launch_configuration_t(tables.num_blocks, tables.block_size, tables.dynamic_shared_memory_size), tables.pieces, tables.blocks);
	}

	launch_configuration_t resolve_launch_configuration(
		device::properties_t           device_properties,
		device_function::attributes_t  kernel_function_attributes,
		size_t                         length,
		size_t                         segment_length,
		launch_configuration_limits_t  limits = { nullopt, nullopt, nullopt }) const
	{
		launch_config_resolution_params_t<IndexSize, Uncompressed, Compressed, UnaryModelFunction> params(
			device_properties, kernel_function_attributes, length, segment_length);
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
	with_typed_parameters(launch_config, arguments, [&](auto... parameters) { launch(stream, launch_config, parameters...); });
}

template <unsigned IndexSize, typename Uncompressed, typename Compressed, typename UnaryModelFunction>
template <typename Sink>
void kernel_t<IndexSize, Uncompressed, Compressed, UnaryModelFunction>::with_typed_parameters(
	const launch_configuration_t&  launch_config,
	const arguments_type&          arguments,
	Sink&&                         sink)
{
	using index_type = uint_t<IndexSize>;

	auto decompressed               = argument<Uncompressed*>(arguments, "decompressed");
	auto compressed_input           = argument<const Compressed*>(arguments, "compressed_input");
	auto segment_model_coefficients = argument<const model_coefficients_type*>(arguments, "segment_model_coefficients");
	auto length                     = argument<index_type>(arguments, "length");
	auto segment_length             = argument<index_type>(arguments, "segment_length");

	// One model record per segment.  The kernel's last parameter exists for signature parity
	// with the reference (frame_of_reference.cu:93-100 computes it the same way).
	auto num_segments = util::div_rounding_up(length, segment_length);
	auto num_blocks = launch_config.grid_dimensions.x;
	index_type segments_per_block = num_blocks ? util::div_rounding_up(num_segments, num_blocks) : 0;

	sink(decompressed, compressed_input, segment_model_coefficients, (size_type) length, (size_type) segment_length, segments_per_block);
}

template <unsigned IndexSize, typename Uncompressed, typename Compressed, typename UnaryModelFunction>
const device_function_t kernel_t<IndexSize, Uncompressed, Compressed, UnaryModelFunction>::get_device_function() const
{
	// (the any-alignment build runs with the configuration resolved for `decompress`; not merged into its attributes)
	return device_function_t(reinterpret_cast<const void*>(decompress<IndexSize, Uncompressed, Compressed, UnaryModelFunction>), { },
		{ reinterpret_cast<const void*>(decompress_at_any_alignment<IndexSize, Uncompressed, Compressed, UnaryModelFunction>) });
}

/// kernel_t<...>::launch as a free function: the one-line replacement for a direct launch of `decompress` ("mode C")
template <unsigned IndexSize, typename Uncompressed, typename Compressed, typename UnaryModelFunction>
inline void launch(stream::id_t stream, const launch_configuration_t& launch_config,
	Uncompressed* decompressed, const Compressed* compressed_input, const typename UnaryModelFunction::coefficients_type* segment_model_coefficients,
	size_type_by_index_size<IndexSize> length, size_type_by_index_size<IndexSize> segment_length, uint_t<IndexSize> num_segments_to_decompress_per_block)
{
	kernel_t<IndexSize, Uncompressed, Compressed, UnaryModelFunction>::launch(stream, launch_config, decompressed, compressed_input, segment_model_coefficients,
		length, segment_length, num_segments_to_decompress_per_block);
}

} // namespace frame_of_reference
} // namespace decompression
} // namespace kernels
} // namespace cuda
