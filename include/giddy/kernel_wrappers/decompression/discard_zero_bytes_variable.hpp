// Kernel wrapper for variable-width discard-zero-bytes decompression.  Argument-map keys and exact
// types follow the reference's src/kernel_wrappers/decompression/discard_zero_bytes_variable.cu:52-95.
#pragma once
#include <limits>
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/decompression/discard_zero_bytes_variable.hpp"
#include "giddy/stream_scratch.hpp"


namespace cuda {
namespace kernels {
namespace decompression {
namespace discard_zero_bytes {
namespace variable_width {

/// output bytes per launch of a long column (see launch(): a launch boundary brings the XCDs' write fronts together again; smaller pieces lose more to launch tails than they gain)
enum : uint64_t { launch_bytes = 1ull << 30 };

/// GIDDY_DZBV_FINE_ANCHORS=0 in the environment: long segments are decoded as they are (measurements)
inline int fine_from_environment()
{
	static const int on = [] { const char* v = std::getenv("GIDDY_DZBV_FINE_ANCHORS"); return (v && v[0] == '0') ? 0 : 1; }();
	return on;
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize = sizeof(native_word_t)>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
	REGISTERED_KERNEL_WRAPPER_TYPED_LAUNCH(kernel_t);
	REGISTERED_KERNEL_WRAPPER_BATCHES(kernel_t);

	using element_size_t     = unsigned;
	using uncompressed_type  = uint_t<UncompressedSize>;
	using element_sizes_container_size_type = uint_t<ElementSizesContainerSize>;
	using size_type          = size_type_by_index_size<IndexSize>;

	launch_configuration_t resolve_launch_configuration(
		device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
		size_t length_in_elements, size_t segment_length, launch_configuration_limits_t limits) const
	{
		launch_config_resolution_params_t<IndexSize, UncompressedSize, ElementSizesContainerSize> params(
			device_properties, kernel_function_attributes, length_in_elements, segment_length);
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}

	/// the argument map unpacked: sink(launch()'s parameters) — for the one launch and for a piece of a batch alike
	template <typename Sink>
	static void with_typed_parameters(const launch_configuration_t&, const arguments_type& arguments, Sink&& sink)
	{
		using index_type = uint_t<IndexSize>;
		auto decompressed                 = argument<uncompressed_type*>(arguments, "decompressed");
		auto compressed_data              = argument<const unsigned char*>(arguments, "compressed_data");
		auto packed_element_sizes         = argument<const element_sizes_container_size_type*>(arguments, "packed_element_sizes");
		auto position_anchors             = argument<const index_type*>(arguments, "position_anchors");
		auto segment_length               = argument<index_type>(arguments, "segment_length");
		auto length                       = argument<index_type>(arguments, "length");
		auto min_represented_element_size = argument<element_size_t>(arguments, "min_represented_element_size");
		auto bits_per_element_size        = argument<unsigned>(arguments, "bits_per_element_size");
		sink(decompressed, compressed_data, packed_element_sizes, position_anchors, (size_type) segment_length, (size_type) length, min_represented_element_size, bits_per_element_size);
	}

	/// a column as one piece of a batch (giddy/batch.hpp): decoded by `decompress` on its segments as they are (pieces are short; launch()'s finer anchors
	/// for long segments and its cuts of long columns are for columns that fill the chip by themselves)
	static void add_piece(batch::table_t& table, const launch_configuration_t& piece_config,
		uncompressed_type* decompressed, const unsigned char* compressed_data, const element_sizes_container_size_type* packed_element_sizes,
		const uint_t<IndexSize>* position_anchors, size_type segment_length, size_type length,
		element_size_t min_represented_element_size, unsigned bits_per_element_size)
	{
		if (segment_length == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
		table.add(piece_config, decompressed, compressed_data, packed_element_sizes, position_anchors, segment_length, length, min_represented_element_size, bits_per_element_size);
	}

	/// ... and the batch's ONE launch
	static void launch_batch(stream::id_t stream, const batch::on_device_t& tables)
	{
		cuda::enqueue_launch(reinterpret_cast<const void*>(decompress_pieces<IndexSize, UncompressedSize, ElementSizesContainerSize>), stream,
			launch_configuration_t(tables.num_blocks, tables.block_size, tables.dynamic_shared_memory_size), tables.pieces, tables.blocks);
	}

	/// THE decode of a variable-width discard-zero-bytes column on `stream`: what the type-erased enqueue_launch(),
	/// cuda::kernel::enqueue_launch(kernel, ...) with typed parameters and variable_width::launch<...>() all run.  `decompress` is the one
	/// decoder; this function gives long segments finer anchors first (scratch from stream_scratch — none to be had: the segments as
	/// they are) and cuts a long column into consecutive launches of about `launch_bytes` of output.
	static void launch(stream::id_t stream, const launch_configuration_t& launch_config,
		uncompressed_type* decompressed, const unsigned char* compressed_data, const element_sizes_container_size_type* packed_element_sizes,
		const uint_t<IndexSize>* position_anchors, size_type segment_length, size_type length,
		element_size_t min_represented_element_size, unsigned bits_per_element_size);
};

template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
launch_configuration_t kernel_t<IndexSize, UncompressedSize, ElementSizesContainerSize>::resolve_launch_configuration(
	device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
	arguments_type extra_arguments, launch_configuration_limits_t limits) const
{
	auto length         = argument<size_t>(extra_arguments, "length");          // overall, in elements
	auto segment_length = argument<size_t>(extra_arguments, "segment_length");  // elements per anchored segment
	return resolve_launch_configuration(device_properties, kernel_function_attributes, length, segment_length, limits);
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
void kernel_t<IndexSize, UncompressedSize, ElementSizesContainerSize>::enqueue_launch_with(
	stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	with_typed_parameters(launch_config, arguments, [&](auto... parameters) { launch(stream, launch_config, parameters...); });
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
void kernel_t<IndexSize, UncompressedSize, ElementSizesContainerSize>::launch(stream::id_t stream, const launch_configuration_t& launch_config,
	uncompressed_type* decompressed, const unsigned char* compressed_data, const element_sizes_container_size_type* packed_element_sizes,
	const uint_t<IndexSize>* position_anchors, size_type segment_length, size_type length,
	element_size_t min_represented_element_size, unsigned bits_per_element_size)
{
	using index_type = uint_t<IndexSize>;
	const void* const base_kernel = reinterpret_cast<const void*>(decompress<IndexSize, UncompressedSize, ElementSizesContainerSize>);
	// A long launch loses its compact write front: workgroup b runs on XCD b % 8, the XCDs work through their eighths of the grid at
	// paces that differ by some 10 %, and once their fronts are tens of MiB apart the memory system serves them at 65-70 % of the HBM peak
	// instead of 81 % (profiles/r03_dzbv_xcd_drift_2p30.txt).  A launch boundary brings the fronts together again, so a long column
	// is decoded as consecutive launches of about `launch_bytes` of output each (2^31 elements: 69 -> 78 %): whole
	// segments, whole size containers and 16-byte aligned outputs per piece; the anchors are absolute, so compressed_data stays put.
	// Long segments: anchors for every 1024 elements from the size fields (kernels/..., namespace fine), in stream-ordered scratch memory,
	// then the decoder on segments of 1024 — where the wide path applies (1-, 2-, 4-bit fields; 4- and 8-byte elements).
	// Segments of a row or more that are NOT whole 1024s (1000, 1023, 10 000 ...: one at a time, ragged ends the slow way — 35-63 % of the HBM peak): every k-th
	// anchor first, k * segment_length the least common multiple with 1024 — the anchors are checkpoints of one running sum (the reference's cuh:366-369) — and
	// then the finer anchors below as for any long segment.  (Shorter than a row: the kernel itself takes every k-th anchor, no launch needed.)
	stream_scratch::lease coarser_anchors(stream);
	{
		constexpr unsigned E = 16 / UncompressedSize, row = 64 * E;
		const uint64_t n = length, L = segment_length;
		const unsigned bits = bits_per_element_size;
		const bool eligible = E <= 4 && L % fine::sub != 0 && detail::wide_path_applies(bits, ElementSizesContainerSize, E)
			&& (reinterpret_cast<uintptr_t>(decompressed) & 15) == 0 && (reinterpret_cast<uintptr_t>(packed_element_sizes) & 15) == 0 && fine_from_environment() != 0;
		if (eligible && n <= L && n > fine::sub && util::round_up(n, (uint64_t) fine::sub) <= (uint64_t) std::numeric_limits<size_type>::max()) {
			segment_length = (size_type) util::round_up(n, (uint64_t) fine::sub);   // (the column is one segment: of any length that holds it — where such a length can be stated: else the caller's)
		}
		else if (eligible && n > L && L >= row) {
			uint64_t common = fine::sub, rest = L;
			while (rest != 0) { const uint64_t t = common % rest; common = rest; rest = t; }   // gcd(1024, L)
			const uint64_t stride = fine::sub / common, coarser_length = L * stride;
			const uint64_t count = util::div_rounding_up(n, coarser_length);
			if (coarser_length <= (uint64_t) std::numeric_limits<size_type>::max() / 2 && coarser_anchors.take(count * sizeof(index_type))) {
				auto* coarser = static_cast<index_type*>(coarser_anchors.memory());
				cuda::enqueue_launch(reinterpret_cast<const void*>(fine::every_kth_anchor<IndexSize>), stream,
					launch_configuration_t((grid_dimension_t) std::min<uint64_t>(util::div_rounding_up(count, (uint64_t) fine::threads), 1u << 16), fine::threads),
					coarser, position_anchors, count, stride);
				position_anchors = coarser;
				segment_length = (size_type) coarser_length;
			}
		}
	}
	stream_scratch::lease fine_anchors(stream);
	{
		constexpr unsigned E = 16 / UncompressedSize;
		const uint64_t n = length, L = segment_length;
		const unsigned bits = bits_per_element_size;
		if (E <= 4 && n > 0 && L > fine::sub && L % fine::sub == 0 && detail::wide_path_applies(bits, ElementSizesContainerSize, E)
			&& (reinterpret_cast<uintptr_t>(decompressed) & 15) == 0 && (reinterpret_cast<uintptr_t>(packed_element_sizes) & 15) == 0 && fine_from_environment() != 0) {
			const uint64_t subs = util::div_rounding_up(n, (uint64_t) fine::sub);
			if (fine_anchors.take((subs + 1) * (sizeof(uint64_t) + sizeof(index_type)))) {
				auto* totals = static_cast<uint64_t*>(fine_anchors.memory());
				auto* anchors = reinterpret_cast<index_type*>(totals + subs + 1);
				const auto* size_words = reinterpret_cast<const uint32_t*>(packed_element_sizes);
				const unsigned min_size = min_represented_element_size;
				auto with_bits = [&](auto bits_constant) {
					constexpr unsigned Bits = decltype(bits_constant)::value;
					if (L / fine::sub <= 64) {   // one launch: a wavefront per segment
						const uint64_t segments = util::div_rounding_up(n, L);
						cuda::enqueue_launch(reinterpret_cast<const void*>(fine::anchors_of_short_segments<Bits, IndexSize>), stream,
							launch_configuration_t((grid_dimension_t) std::min<uint64_t>(util::div_rounding_up(subs, (uint64_t) 64 * fine::threads / 64 / 2), 1u << 16), fine::threads),
							anchors, size_words, position_anchors, n, L, min_size);
						return;
					}
					cuda::enqueue_launch(reinterpret_cast<const void*>(fine::totals<Bits>), stream,
						launch_configuration_t((grid_dimension_t) std::min<uint64_t>(util::div_rounding_up(subs, (uint64_t) 64 * fine::threads / 64), 1u << 16), fine::threads),
						totals, size_words, n, min_size);
					const uint64_t blocks = util::div_rounding_up(n, L) * util::div_rounding_up(L / fine::sub, (uint64_t) fine::subs_per_block);
					cuda::enqueue_launch(reinterpret_cast<const void*>(fine::anchors<IndexSize>), stream,
						launch_configuration_t((grid_dimension_t) std::min<uint64_t>(blocks, 1u << 16), fine::threads),
						anchors, static_cast<const uint64_t*>(totals), position_anchors, n, L);
				};
				if (bits == 2) { with_bits(std::integral_constant<unsigned, 2>()); }
				else if (bits == 4) { with_bits(std::integral_constant<unsigned, 4>()); }
				else { with_bits(std::integral_constant<unsigned, 1>()); }
				position_anchors = anchors;
				segment_length = (index_type) fine::sub;
			}
		}
	}
	const size_t n = length, L = segment_length;
	const size_t per = bits_per_element_size == 0 ? 1 : (8 * ElementSizesContainerSize) / bits_per_element_size;
	const size_t segments = L == 0 ? 0 : util::div_rounding_up(n, L);
	const size_t pieces = std::max<size_t>(1, (n * UncompressedSize + (size_t) launch_bytes / 2) / (size_t) launch_bytes);
	const size_t granule = std::max<size_t>(per, 1) * 16;
	const size_t segments_per_piece = pieces <= 1 ? segments : util::round_up(util::div_rounding_up(segments, pieces), granule);
	for (size_t first = 0; first < std::max<size_t>(segments, 1); first += std::max<size_t>(segments_per_piece, 1)) {
		const size_t begin = first * L, end = std::min(n, (first + segments_per_piece) * L);
		auto piece_config = launch_config;
		const bool with_fine_anchors = fine_anchors.memory() != nullptr;   // (then the caller's grid was resolved for other segments than the launch's)
		if (begin != 0 || end != n || with_fine_anchors) {
			// a piece's own covering grid, within the grid the caller resolved for the column (which may be capped, or limited by the caller)
			const size_t chunks = util::div_rounding_up(util::div_rounding_up(end - begin, L), (size_t) detail::segments_per_wave_chunk(L, 64 * (16 / UncompressedSize)));
			const size_t wavefronts_per_block = std::max<size_t>(1, launch_config.block_dimensions.x / warp_size);
			const size_t covering = util::div_rounding_up(chunks, std::min<size_t>(wavefronts_per_block, detail::max_waves_per_block));
			piece_config.grid_dimensions.x = (grid_dimension_t) std::max<size_t>(1, std::min<size_t>(covering, with_fine_anchors ? (size_t) 1 << 18 : (size_t) launch_config.grid_dimensions.x));
		}
		cuda::enqueue_launch(base_kernel, stream, piece_config,
			decompressed + begin, compressed_data, packed_element_sizes + begin / per, position_anchors + first,
			(size_type) segment_length, (size_type) (end - begin), min_represented_element_size, bits_per_element_size);
		if (segments_per_piece == 0) { break; }
	}
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
const device_function_t kernel_t<IndexSize, UncompressedSize, ElementSizesContainerSize>::get_device_function() const
{
	return reinterpret_cast<const void*>(decompress<IndexSize, UncompressedSize, ElementSizesContainerSize>);
}

/// kernel_t<...>::launch as a free function: the one-line replacement for a direct launch of `decompress` ("mode C")
template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize = sizeof(native_word_t)>
inline void launch(stream::id_t stream, const launch_configuration_t& launch_config,
	uint_t<UncompressedSize>* decompressed, const unsigned char* compressed_data, const uint_t<ElementSizesContainerSize>* packed_element_sizes,
	const uint_t<IndexSize>* position_anchors, size_type_by_index_size<IndexSize> segment_length, size_type_by_index_size<IndexSize> length,
	unsigned min_represented_element_size, unsigned bits_per_element_size)
{
	kernel_t<IndexSize, UncompressedSize, ElementSizesContainerSize>::launch(stream, launch_config, decompressed, compressed_data, packed_element_sizes,
		position_anchors, segment_length, length, min_represented_element_size, bits_per_element_size);
}

} // namespace variable_width
} // namespace discard_zero_bytes
} // namespace decompression
} // namespace kernels
} // namespace cuda
