// Kernel wrapper for Delta decompression.  Argument-map keys and exact types follow the
// reference's src/kernel_wrappers/decompression/delta.cu:49-83.
#pragma once
#include <mutex>

#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/functors/functors.hpp"
#include "giddy/kernels/decompression/delta.hpp"
#include "giddy/kernels/reduction/scan.hpp"
#include "giddy/stream_scratch.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace delta {

namespace detail {

enum : unsigned { long_segment_from = 1u << 17, longest_part = 1u << 16, shortest_part = 1u << 12 };
enum : unsigned { chained_from_rows = 128 };   // segments of at least this many rows go by chained tiles (fewer: the wave and block paths)

/// part_anchors[p] (holding the sum of part p's deltas on entry) = the value before part p: the segment's anchor + the sums of the
/// segment's earlier parts.  A thread per segment (long segments are few).
template <unsigned UncompressedSize>
__global__ void anchors_of_parts(uint_t<UncompressedSize>* __restrict__ part_anchors, const uint_t<UncompressedSize>* __restrict__ anchor_values,
	uint64_t num_segments, uint64_t parts_per_segment, uint64_t num_parts)
{
	for (uint64_t s = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; s < num_segments; s += (uint64_t) gridDim.x * blockDim.x) {
		uint_t<UncompressedSize> running = anchor_values[s];
		for (uint64_t p = s * parts_per_segment; p < (s + 1) * parts_per_segment && p < num_parts; p++) {
			const uint_t<UncompressedSize> sum = part_anchors[p];
			part_anchors[p] = running;
			running = (uint_t<UncompressedSize>) (running + sum);
		}
	}
}

using cuda::current_device_properties;   // (kernel_wrappers/common.hpp)

/// switches read from the environment ONCE per process (not on the launch path)
struct environment {
	bool chained_from_given = false;
	uint64_t chained_from = 0;
	static const environment& get()
	{
		static const environment e = [] {
			environment read;
			if (const char* v = std::getenv("GIDDY_DELTA_CHAINED_FROM")) { read.chained_from_given = true; read.chained_from = (uint64_t) std::strtoull(v, nullptr, 10); }
			return read;
		}();
		return e;
	}
};

} // namespace detail

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
	REGISTERED_KERNEL_WRAPPER_TYPED_LAUNCH(kernel_t);
	REGISTERED_KERNEL_WRAPPER_BATCHES(kernel_t);

	using compressed_type   = uint_t<CompressedSize>;
	using uncompressed_type = uint_t<UncompressedSize>;
	using size_type         = size_type_by_index_size<IndexSize>;

	launch_configuration_t resolve_launch_configuration(
		device::properties_t           device_properties,
		device_function::attributes_t  kernel_function_attributes,
		size_t                         length,
		size_t                         segment_length,
		launch_configuration_limits_t  limits) const
	{
		launch_config_resolution_params_t<IndexSize, UncompressedSize, CompressedSize> params(
			device_properties, kernel_function_attributes, length, segment_length, limits.dynamic_shared_memory);
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}

	/// the argument map unpacked: sink(launch()'s parameters) — for the one launch and for a piece of a batch alike
	template <typename Sink>
	static void with_typed_parameters(const launch_configuration_t&, const arguments_type& arguments, Sink&& sink)
	{
		auto decompressed      = argument<uncompressed_type*>(arguments, "decompressed");
		auto compressed_input  = argument<const compressed_type*>(arguments, "compressed_input");
		auto anchor_values     = argument<const uncompressed_type*>(arguments, "anchor_values");
		auto length            = argument<uint_t<IndexSize>>(arguments, "length");
		auto segment_length    = argument<uint_t<IndexSize>>(arguments, "segment_length");
		// (an extension: ceil(length / 4096) + 1024 entries, for columns of very long segments where no stream scratch is to be had)
		auto part_anchors      = optional_argument<uncompressed_type*>(arguments, "part_anchors_scratch");
		sink(decompressed, compressed_input, anchor_values, (size_type) length, (size_type) segment_length, part_anchors ? part_anchors.value() : nullptr);
	}

	/// a column as one piece of a batch (giddy/batch.hpp): decoded by `decompress` — a wavefront or a workgroup per segment, whatever the segments' length; pieces are
	/// short, and launch()'s chained tiles are for segments that leave the chip empty by themselves (`part_anchors_scratch` is not used)
	static void add_piece(batch::table_t& table, const launch_configuration_t& piece_config,
		uncompressed_type* decompressed, const compressed_type* compressed_input, const uncompressed_type* anchor_values,
		size_type length, size_type segment_length, uncompressed_type* = nullptr)
	{
		if (segment_length == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
		table.add(piece_config, decompressed, compressed_input, anchor_values, length, segment_length);
	}

	/// ... and the batch's ONE launch
	static void launch_batch(stream::id_t stream, const batch::on_device_t& tables)
	{
		cuda::enqueue_launch(reinterpret_cast<const void*>(batch::decode_pieces<piece_body<IndexSize, UncompressedSize, CompressedSize>>), stream,
			launch_configuration_t(tables.num_blocks, tables.block_size, tables.dynamic_shared_memory_size), tables.pieces, tables.blocks);
	}

	/// THE decode of a Delta column on `stream`: what the type-erased enqueue_launch(), cuda::kernel::enqueue_launch(kernel, ...) with
	/// typed parameters and delta::launch<...>() all run.  Picks among the kernels of kernels/decompression/delta.hpp by the column's shape:
	/// `decompress` (a wavefront or a workgroup per segment, launched with `launch_config`), chained tiles for long segments (a
	/// configuration of their own; per-tile states in stream_scratch), or — given `part_anchors_scratch` — long segments in parts.
	static void launch(
		stream::id_t                   stream,
		const launch_configuration_t&  launch_config,
		uncompressed_type*             decompressed,
		const compressed_type*         compressed_input,
		const uncompressed_type*       anchor_values,
		size_type                      length,
		size_type                      segment_length,
		uncompressed_type*             part_anchors_scratch = nullptr);
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
	return resolve_launch_configuration(device_properties, kernel_function_attributes, length, segment_length, limits);
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
void kernel_t<IndexSize, UncompressedSize, CompressedSize>::enqueue_launch_with(
	stream::id_t                   stream,
	const launch_configuration_t&  launch_config,
	const arguments_type&                 arguments) const
{
	with_typed_parameters(launch_config, arguments, [&](auto... parameters) { launch(stream, launch_config, parameters...); });
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
void kernel_t<IndexSize, UncompressedSize, CompressedSize>::launch(
	stream::id_t                   stream,
	const launch_configuration_t&  launch_config,
	uncompressed_type*             decompressed,
	const compressed_type*         compressed_input,
	const uncompressed_type*       anchor_values,
	size_type                      length,
	size_type                      segment_length,
	uncompressed_type*             part_anchors)
{
	const void* const base_kernel = reinterpret_cast<const void*>(decompress<IndexSize, UncompressedSize, CompressedSize>);

	// A column at addresses aligned for its elements only (a slice decoded into `decompressed + a`): the build that takes it by windows whose rows follow the output's
	// boundaries (kernels/decompression/delta.hpp, decompress_at_any_alignment) — segments of up to 32 rows, the wave paths' workgroups
	{
		using traits = detail::row_traits<UncompressedSize, CompressedSize>;
		if ((!memory_access::is_aligned(decompressed, 16) || !memory_access::is_aligned(compressed_input, traits::E * CompressedSize))
			&& length > 0 && segment_length > 0 && (uint64_t) segment_length <= (uint64_t) detail::rows_per_chunk * traits::row
			&& launch_config.block_dimensions.x <= detail::any_length_waves * warp_size && launch_config.block_dimensions.y == 1 && launch_config.block_dimensions.z == 1) {
			cuda::enqueue_launch(reinterpret_cast<const void*>(decompress_at_any_alignment<IndexSize, UncompressedSize, CompressedSize>), stream, launch_config,
				decompressed, compressed_input, anchor_values, length, segment_length);
			return;
		}
		// ... and LONGER segments at such addresses (round 6): the windows' bases from a pre-pass over the deltas — two small launches and scratch for two values per
		// window —, then the same windows (kernels/decompression/delta.hpp, decompress_windows_at_any_alignment): 17-21 % of the HBM peak element by element -> 51-64 %,
		// profiles/r06_unaligned_bases.txt.  No scratch to be had (or segments of more than 2^16 windows: the pre-pass adds a segment's windows up per window): element by element.
		constexpr uint64_t window = (uint64_t) detail::rows_per_chunk * traits::row;
		if ((!memory_access::is_aligned(decompressed, 16) || !memory_access::is_aligned(compressed_input, traits::E * CompressedSize))
			&& length > 0 && (uint64_t) segment_length > window && (uint64_t) segment_length / window <= (1u << 16)) {
			const uint64_t windows = util::div_rounding_up((uint64_t) length, window);
			stream_scratch::lease scratch(stream);
			if (windows <= 0x7FFFFFFFull && scratch.take(2 * windows * sizeof(uncompressed_type))) {
				auto* tail_sums = static_cast<uncompressed_type*>(scratch.memory());
				auto* window_bases = tail_sums + windows;
				const launch_configuration_t per_window((grid_dimension_t) util::div_rounding_up(windows, (uint64_t) detail::any_length_waves), detail::any_length_waves * warp_size);
				cuda::enqueue_launch(reinterpret_cast<const void*>(window_tail_sums<IndexSize, UncompressedSize, CompressedSize>), stream, per_window,
					tail_sums, compressed_input, length, segment_length);
				cuda::enqueue_launch(reinterpret_cast<const void*>(bases_of_windows<IndexSize, UncompressedSize, CompressedSize>), stream, per_window,
					window_bases, static_cast<const uncompressed_type*>(tail_sums), anchor_values, length, segment_length);
				cuda::enqueue_launch(reinterpret_cast<const void*>(decompress_windows_at_any_alignment<IndexSize, UncompressedSize, CompressedSize>), stream, per_window,
					decompressed, compressed_input, anchor_values, length, segment_length, static_cast<const uncompressed_type*>(window_bases));
				return;
			}
		}
	}

	// Long segments — more rows than a wavefront takes by itself — as tiles of their own, chained by a decoupled look-back through
	// per-tile states in stream-ordered scratch memory (kernels/decompression/delta.hpp, namespace chained): every tile a workgroup
	// whatever the segment length, the deltas read once.  (GIDDY_DELTA_CHAINED_FROM in the environment: the segment length, in elements,
	// from which this path is taken; 0 turns it off.)
	{
		using traits = detail::row_traits<UncompressedSize, CompressedSize>;
		const uint64_t n = length, L = segment_length;
		const auto& environment = detail::environment::get();
		// (64-bit elements from 4-byte deltas: from one tile — 64 rows — on: the block path is at 50-51 % for them at every length, tiles at 71-75 %)
		const uint64_t chained_from = environment.chained_from_given ? environment.chained_from
			: (UncompressedSize == 8 && CompressedSize == 4) ? chained::tile_elements<UncompressedSize, CompressedSize, 4>() : (uint64_t) detail::chained_from_rows * traits::row;
		const bool aligned = memory_access::is_aligned(decompressed, 16) && memory_access::is_aligned(compressed_input, 16);
		const unsigned waves = chained::waves_per_tile_for<UncompressedSize, CompressedSize>(n, L);
		// ... where they beat the block path (a workgroup of up to 16 wavefronts per segment): always when there are fewer segments than compute units,
		// and for 64-bit values from 1-2-byte deltas from one tile per segment on (<4,8,2>, 2^28 elements, segments of 2^14 ...
		// 2^20: 76 / 76 / 75 / 77 % against 75 / 70 / 69 / 66 %) — and from 4-byte deltas (round 4: <8,8,4>, 2^13 ... 2^18: 71-75 % against 50-51 %);
		// 32-bit values keep the block path while it has enough segments (<4,4,1>: 88 % against 77-83 %)
		const uint64_t num_segments = L == 0 ? 0 : util::div_rounding_up(n, L);
		const bool preferred = num_segments < (uint64_t) detail::current_device_properties().multiProcessorCount || UncompressedSize == 8;
		const bool forced = environment.chained_from_given;
		if (chained_from != 0 && n > 0 && L >= chained_from && (preferred || forced) && L > (uint64_t) detail::rows_per_chunk * traits::row && aligned && waves != 0) {
			using state_type = chained::tile_state<UncompressedSize>;
			auto launch_tiles = [&](auto waves_constant) {
				constexpr unsigned Waves = decltype(waves_constant)::value;
				const uint64_t tiles = util::div_rounding_up(n, chained::tile_elements<UncompressedSize, CompressedSize, Waves>());
				// (a workgroup per tile, exactly: a grid stride over more workgroups than are resident would leave resident ones waiting for tiles of
				// workgroups that cannot start — the kernel's bounded spin then falls back to summing the deltas itself: correct, and hopeless)
				if (tiles > 0x7FFFFFFFull) { return false; }
				stream_scratch::lease scratch(stream);
				if (!scratch.take_zeroed(tiles * sizeof(state_type))) { return false; }
				auto* states = static_cast<state_type*>(scratch.memory());
				launch_configuration_t tiles_config((grid_dimension_t) tiles, Waves * warp_size);
				cuda::enqueue_launch(reinterpret_cast<const void*>(chained::decompress<IndexSize, UncompressedSize, CompressedSize, Waves>), stream, tiles_config,
					decompressed, compressed_input, anchor_values, length, segment_length, states);
				return true;
			};
			if (waves == 16 ? launch_tiles(std::integral_constant<unsigned, 16>()) : waves == 8 ? launch_tiles(std::integral_constant<unsigned, 8>())
				: launch_tiles(std::integral_constant<unsigned, 4>())) { return; }
		}
	}

	// An extension for columns with very long segments (a single un-anchored column is ONE segment: one workgroup's work, 8 % of the
	// peak at 2^24 elements): given "part_anchors_scratch" — uncompressed_type*, ceil(length / 4096) + 1024 entries — segments of at least 2^17
	// elements are decoded in PARTS: the parts' delta sums (the two-phase scan's first kernel), the value before each part, and
	// then this decoder with the parts as its segments — when there are fewer such segments than compute units.  The deltas are read twice:
	// 55 % of the peak instead of 8 % for segments of 2^24 elements, 33 % instead of 0.5 % for one segment of 2^28.
	if (part_anchors != nullptr && segment_length >= detail::long_segment_from && length > 0) {
		uint64_t part = detail::longest_part;
		while (part >= detail::shortest_part && segment_length % part != 0) { part >>= 1; }
		const auto& properties = detail::current_device_properties();
		const uint64_t num_segments = util::div_rounding_up((uint64_t) length, (uint64_t) segment_length);
		// (a workgroup sweeps a segment: with a segment per compute unit or more the ordinary path is the faster one — 66 against 59 %)
		if (part >= detail::shortest_part && num_segments < (uint64_t) properties.multiProcessorCount) {
			namespace scan = cuda::kernels::reduction::scan;
			using plus = functors::plus<uncompressed_type>;
			using widen = functors::cast<compressed_type, uncompressed_type>;
			const uint64_t num_parts = util::div_rounding_up((uint64_t) length, part);
			scan::reduce_segments::launch_config_resolution_params_t<IndexSize, plus, compressed_type, widen> reduction(properties, length, part);
			auto reduction_config = cuda::kernels::resolve_launch_configuration(reduction);
			if (reduction_config.grid_dimensions.x > num_parts) { reduction_config.grid_dimensions.x = (grid_dimension_t) num_parts; }
			cuda::enqueue_launch(reinterpret_cast<const void*>(scan::reduce_segments::reduce_segments<IndexSize, plus, compressed_type, widen>), stream, reduction_config,
				part_anchors, compressed_input, length, (size_type) part);
			cuda::enqueue_launch(reinterpret_cast<const void*>(detail::anchors_of_parts<UncompressedSize>), stream,
				launch_configuration_t((grid_dimension_t) util::div_rounding_up(num_segments, (uint64_t) 64), 64),
				part_anchors, anchor_values, num_segments, (uint64_t) segment_length / part, num_parts);
			launch_config_resolution_params_t<IndexSize, UncompressedSize, CompressedSize> by_parts(properties, device_function_t(base_kernel).attributes(), length, part);
			cuda::enqueue_launch(base_kernel, stream, cuda::kernels::resolve_launch_configuration(by_parts),
				decompressed, compressed_input, static_cast<const uncompressed_type*>(part_anchors), length, (size_type) part);
			return;
		}
	}
	cuda::enqueue_launch(base_kernel, stream, launch_config, decompressed, compressed_input, anchor_values, length, segment_length);
}

/// The kernel of the reference's signature (what a direct launch — the reference's third mode of use — runs: right for every segment
/// length, a workgroup per long segment).  Chained tiles are launched with a configuration of their own, so the attributes of this
/// kernel are the ones the caller's configuration is resolved against.
template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
const device_function_t kernel_t<IndexSize, UncompressedSize, CompressedSize>::get_device_function() const
{
	// (the any-alignment build runs with the configuration resolved for `decompress`; not merged into its attributes)
	return device_function_t(reinterpret_cast<const void*>(decompress<IndexSize, UncompressedSize, CompressedSize>), { },
		{ reinterpret_cast<const void*>(decompress_at_any_alignment<IndexSize, UncompressedSize, CompressedSize>) });
}

/// kernel_t<...>::launch as a free function: the one-line replacement for a direct launch of `decompress` ("mode C")
template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
inline void launch(stream::id_t stream, const launch_configuration_t& launch_config,
	uint_t<UncompressedSize>* decompressed, const uint_t<CompressedSize>* compressed_input, const uint_t<UncompressedSize>* anchor_values,
	size_type_by_index_size<IndexSize> length, size_type_by_index_size<IndexSize> segment_length, uint_t<UncompressedSize>* part_anchors_scratch = nullptr)
{
	kernel_t<IndexSize, UncompressedSize, CompressedSize>::launch(stream, launch_config, decompressed, compressed_input, anchor_values, length, segment_length, part_anchors_scratch);
}

} // namespace delta
} // namespace decompression
} // namespace kernels
} // namespace cuda
