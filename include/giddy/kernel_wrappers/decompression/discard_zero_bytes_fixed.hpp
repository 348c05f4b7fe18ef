// Kernel wrapper for fixed-width discard-zero-bytes decompression.  Argument-map keys and exact
// types follow the reference's src/kernel_wrappers/decompression/discard_zero_bytes_fixed.cu:60-97;
// the two defaulted template arguments are part of the factory key.
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/decompression/discard_zero_bytes_fixed.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace discard_zero_bytes {
namespace fixed_width {

using util::endianness_t;

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize,
	endianness_t UncompressedEndianness = util::compilation_target_endianness(),
	serialization_factor_t SerializationFactor = DefaultSerializationFactor>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
	REGISTERED_KERNEL_WRAPPER_TYPED_LAUNCH(kernel_t);
	REGISTERED_KERNEL_WRAPPER_BATCHES(kernel_t);

	using uncompressed_type = uint_t<UncompressedSize>;
	using compressed_type   = uint_t<CompressedSize>;
	using size_type         = size_type_by_index_size<IndexSize>;

	/// THE decode on `stream` — what the type-erased enqueue_launch() and cuda::kernel::enqueue_launch(kernel, typed parameters...) run: `decompress`, or its build
	/// for columns at any element offset when the addresses ask for it (kernels/decompression/discard_zero_bytes_fixed.hpp)
	static void launch(stream::id_t stream, const launch_configuration_t& launch_config,
		uncompressed_type* decompressed, const compressed_type* compressed_input, size_type length)
	{
		const void* kernel = stream_tiles::wants_any_alignment_build(decompressed, compressed_input)
			? reinterpret_cast<const void*>(decompress_at_any_alignment<IndexSize, UncompressedSize, CompressedSize, translate(UncompressedEndianness), SerializationFactor>)
			: reinterpret_cast<const void*>(decompress<IndexSize, UncompressedSize, CompressedSize, translate(UncompressedEndianness), SerializationFactor>);
		cuda::enqueue_launch(kernel, stream, launch_config, decompressed, compressed_input, length);
	}

	/// the argument map unpacked: sink(launch()'s parameters) — for the one launch and for a piece of a batch alike
	template <typename Sink>
	static void with_typed_parameters(const launch_configuration_t&, const arguments_type& arguments, Sink&& sink)
	{
		auto decompressed     = argument<uncompressed_type*>(arguments, "decompressed");
		auto compressed_input = argument<const compressed_type*>(arguments, "compressed_input");
		auto length           = argument<uint_t<IndexSize>>(arguments, "length");
		sink(decompressed, compressed_input, (size_type) length);
	}

	/// a column as one piece of a batch (giddy/batch.hpp); hints[3]: some piece lies where `decompress` would go element by element
	static void add_piece(batch::table_t& table, const launch_configuration_t& piece_config,
		uncompressed_type* decompressed, const compressed_type* compressed_input, size_type length)
	{
		if (stream_tiles::wants_any_alignment_build(decompressed, compressed_input)) { table.hints[3] = 1; }
		table.add(piece_config, decompressed, compressed_input, length);
	}

	/// ... and the batch's ONE launch
	static void launch_batch(stream::id_t stream, const batch::on_device_t& tables)
	{
		const void* kernel = tables.hints[3] != 0
			? reinterpret_cast<const void*>(batch::decode_pieces<piece_body<IndexSize, UncompressedSize, CompressedSize, translate(UncompressedEndianness), true>>)
			: reinterpret_cast<const void*>(batch::decode_pieces<piece_body<IndexSize, UncompressedSize, CompressedSize, translate(UncompressedEndianness), false>>);
		cuda::enqueue_launch(kernel, stream, launch_configuration_t(tables.num_blocks, tables.block_size, tables.dynamic_shared_memory_size), tables.pieces, tables.blocks);
	}

	launch_configuration_t resolve_launch_configuration(
		device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
		size_t length, launch_configuration_limits_t limits) const
	{
		launch_config_resolution_params_t<IndexSize, UncompressedSize, CompressedSize, translate(UncompressedEndianness), SerializationFactor>
			params(device_properties, kernel_function_attributes, length);
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}
};

#define GIDDY_DZBF_TEMPLATE template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize, endianness_t UncompressedEndianness, serialization_factor_t SerializationFactor>
#define GIDDY_DZBF_KERNEL kernel_t<IndexSize, UncompressedSize, CompressedSize, UncompressedEndianness, SerializationFactor>

GIDDY_DZBF_TEMPLATE
launch_configuration_t GIDDY_DZBF_KERNEL::resolve_launch_configuration(
	device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
	arguments_type extra_arguments, launch_configuration_limits_t limits) const
{
	auto length = argument<size_t>(extra_arguments, "length");
	return resolve_launch_configuration(device_properties, kernel_function_attributes, length, limits);
}

GIDDY_DZBF_TEMPLATE
void GIDDY_DZBF_KERNEL::enqueue_launch_with(stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	with_typed_parameters(launch_config, arguments, [&](auto... parameters) { launch(stream, launch_config, parameters...); });
}

GIDDY_DZBF_TEMPLATE
const device_function_t GIDDY_DZBF_KERNEL::get_device_function() const
{
	return device_function_t(reinterpret_cast<const void*>(
		decompress<IndexSize, UncompressedSize, CompressedSize, translate(UncompressedEndianness), SerializationFactor>), { },
		{ reinterpret_cast<const void*>(decompress_at_any_alignment<IndexSize, UncompressedSize, CompressedSize, translate(UncompressedEndianness), SerializationFactor>) });
}

#undef GIDDY_DZBF_TEMPLATE
#undef GIDDY_DZBF_KERNEL

} // namespace fixed_width
} // namespace discard_zero_bytes
} // namespace decompression
} // namespace kernels
} // namespace cuda
