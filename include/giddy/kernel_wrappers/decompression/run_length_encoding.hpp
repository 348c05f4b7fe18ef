// Kernel wrapper for run-length decoding.  Argument-map keys and exact types follow the reference's
// src/kernel_wrappers/decompression/run_length_encoding.cu:56-93.
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/decompression/run_length_encoding.hpp"
#include "giddy/kernels/decompression/run_anchors.hpp"
#include "giddy/stream_scratch.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace run_length_encoding {

template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize = UncompressedIndexSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
	REGISTERED_KERNEL_WRAPPER_TYPED_LAUNCH(kernel_t);
	REGISTERED_KERNEL_WRAPPER_BATCHES(kernel_t);

	using run_index_type                = uint_t<RunIndexSize>;
	using uncompressed_index_type       = uint_t<UncompressedIndexSize>;
	using uncompressed_type             = uint_t<UncompressedSize>;
	using run_length_type               = uint_t<RunLengthSize>;
	using uncompressed_index_size_type  = size_type_by_index_size<UncompressedIndexSize>;
	using run_index_size_type           = size_type_by_index_size<RunIndexSize>;
	using size_type                     = uncompressed_index_size_type;

	static_assert(util::is_power_of_2(UncompressedIndexSize), "UncompressedIndexSize is not a power of 2");
	static_assert(util::is_power_of_2(RunIndexSize), "RunIndexSize is not a power of 2");
	static_assert(util::is_power_of_2(RunLengthSize), "RunLengthSize is not a power of 2");

	launch_configuration_t resolve_launch_configuration(
		device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
		size_t uncompressed_length, size_t segment_length, launch_configuration_limits_t limits) const
	{
		launch_config_resolution_params_t<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize> params(
			device_properties, kernel_function_attributes, uncompressed_length, segment_length);
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}

	/// THE decode of a run-length encoded column on `stream`: what the type-erased enqueue_launch(), cuda::kernel::enqueue_launch(kernel, ...)
	/// with typed parameters and run_length_encoding::launch<...>() all run.  Picks the build of the decoder by the column's mean run
	/// (`decompress_sparse_runs` / `decompress_dense_runs` / `decompress`: kernels/decompression/run_length_encoding.hpp) and, for long
	/// segments, finds finer anchors first (run_anchors.hpp; scratch from stream_scratch — none to be had: the segments as they are).
	static void launch(stream::id_t stream, const launch_configuration_t& launch_config,
		uncompressed_type* decompressed, const uncompressed_type* run_data, const run_length_type* run_lengths,
		const run_index_type* position_anchors, const run_length_type* intra_run_anchor_offsets,
		size_type segment_length, size_type num_segments, run_index_size_type num_element_runs, size_type uncompressed_length);

	/// the argument map unpacked: sink(launch()'s parameters) — for the one launch and for a piece of a batch alike
	template <typename Sink>
	static void with_typed_parameters(const launch_configuration_t&, const arguments_type& arguments, Sink&& sink)
	{
		auto decompressed             = argument<uncompressed_type*>(arguments, "decompressed");
		auto run_data                 = argument<const uncompressed_type*>(arguments, "run_data");
		auto run_lengths              = argument<const run_length_type*>(arguments, "run_lengths");
		auto position_anchors         = argument<const run_index_type*>(arguments, "position_anchors");
		auto intra_run_anchor_offsets = argument<const run_length_type*>(arguments, "intra_run_anchor_offsets");
		auto segment_length           = argument<size_type>(arguments, "segment_length");
		auto num_segments             = argument<size_type>(arguments, "num_segments");
		auto num_element_runs         = argument<run_index_size_type>(arguments, "num_element_runs");
		auto uncompressed_length      = argument<size_type>(arguments, "length");
		sink(decompressed, run_data, run_lengths, position_anchors, intra_run_anchor_offsets, segment_length, num_segments, num_element_runs, uncompressed_length);
	}

	/// a column as one piece of a batch (giddy/batch.hpp).  hints: [0] outputs and [1] runs of all pieces — the batch's mean run picks the build of the
	/// decoder, as launch() does for one column —, [2] some piece's run arrays are not 16-byte aligned (no dense build then)
	static void add_piece(batch::table_t& table, const launch_configuration_t& piece_config,
		uncompressed_type* decompressed, const uncompressed_type* run_data, const run_length_type* run_lengths,
		const run_index_type* position_anchors, const run_length_type* intra_run_anchor_offsets,
		size_type segment_length, size_type num_segments, run_index_size_type num_element_runs, size_type uncompressed_length)
	{
		if (segment_length == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
		table.hints[0] += uncompressed_length; table.hints[1] += num_element_runs;
		if (!memory_access::is_aligned(run_lengths, 16) || !memory_access::is_aligned(run_data, 16)) { table.hints[2] = 1; }
		table.add(piece_config, decompressed, run_data, run_lengths, position_anchors, intra_run_anchor_offsets, segment_length, num_segments, num_element_runs, uncompressed_length);
	}

	/// ... and the batch's ONE launch
	static void launch_batch(stream::id_t stream, const batch::on_device_t& tables)
	{
		constexpr bool has_groups = run_expansion::run_groups<UncompressedSize, RunLengthSize>::enabled;
		const uint64_t outputs = tables.hints[0], runs = tables.hints[1];
		const bool sparse = has_groups && runs != 0 && outputs / runs >= 16;
		const bool dense = has_groups && runs != 0 && !sparse && tables.hints[2] == 0;
		const void* kernel = reinterpret_cast<const void*>(batch::decode_pieces<piece_body<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize, false, false>>);
		if constexpr (has_groups) {   // (the other instantiations have the one build)
			if (dense) { kernel = reinterpret_cast<const void*>(batch::decode_pieces<piece_body<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize, true, true>>); }
			else if (!sparse) { kernel = reinterpret_cast<const void*>(batch::decode_pieces<piece_body<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize, true, false>>); }
		}
		// (the dense build wants all the workgroups a CU holds: launch())
		cuda::enqueue_launch(kernel, stream, launch_configuration_t(tables.num_blocks, tables.block_size, dense ? 0 : tables.dynamic_shared_memory_size), tables.pieces, tables.blocks);
	}

	/// the three builds of the decoder, any of which launch() may run with the caller's configuration
	static const void* base_kernel()   { return reinterpret_cast<const void*>(decompress<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize>); }
	static const void* sparse_kernel() { return reinterpret_cast<const void*>(decompress_sparse_runs<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize>); }
	static const void* dense_kernel()  { return reinterpret_cast<const void*>(decompress_dense_runs<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize>); }
	/// ... and, launched with a grid of its own (a workgroup per anchored segment), pre-pass and decoder of long segments in one
	static const void* long_segments_kernel() { return reinterpret_cast<const void*>(decompress_long_segments<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize>); }
};

#define GIDDY_RLE_TEMPLATE template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize>
#define GIDDY_RLE_KERNEL kernel_t<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize>

GIDDY_RLE_TEMPLATE
launch_configuration_t GIDDY_RLE_KERNEL::resolve_launch_configuration(
	device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
	arguments_type extra_arguments, launch_configuration_limits_t limits) const
{
	auto uncompressed_length = argument<size_t>(extra_arguments, "length");
	auto segment_length      = argument<size_t>(extra_arguments, "segment_length");
	return resolve_launch_configuration(device_properties, kernel_function_attributes, uncompressed_length, segment_length, limits);
}

GIDDY_RLE_TEMPLATE
void GIDDY_RLE_KERNEL::enqueue_launch_with(stream::id_t stream, const launch_configuration_t& resolved_launch_config, const arguments_type& arguments) const
{
	with_typed_parameters(resolved_launch_config, arguments, [&](auto... parameters) { launch(stream, resolved_launch_config, parameters...); });
}

GIDDY_RLE_TEMPLATE
void GIDDY_RLE_KERNEL::launch(stream::id_t stream, const launch_configuration_t& resolved_launch_config,
	uncompressed_type* decompressed, const uncompressed_type* run_data, const run_length_type* run_lengths,
	const run_index_type* position_anchors, const run_length_type* intra_run_anchor_offsets,
	size_type segment_length, size_type num_segments, run_index_size_type num_element_runs, size_type uncompressed_length)
{
	// Long segments: entry points for every `fine::sub`-th output first — sums of every 64 run lengths, scanned, searched per finer
	// anchor —, then the decoder on segments of fine::sub (kernels/decompression/run_anchors.hpp).
	namespace fine = run_expansion::fine;
	const uint64_t n = uncompressed_length, L = segment_length, runs = num_element_runs;
	// mean runs of 16 or more: the kernel without the groups-of-runs path (kernels/decompression/run_length_encoding.hpp)
	const bool sparse = run_expansion::run_groups<UncompressedSize, RunLengthSize>::enabled && runs != 0 && n / runs >= 16;
	const bool dense = run_expansion::run_groups<UncompressedSize, RunLengthSize>::enabled && runs != 0 && !sparse
		&& memory_access::is_aligned(run_lengths, 16) && memory_access::is_aligned(run_data, 16);
	const void* const decoder = sparse ? sparse_kernel() : dense ? dense_kernel() : base_kernel();
	// (the dense kernel wants all the workgroups a CU holds: the launch parameters' cap of 6 — LDS reserved for nothing, measured on mean runs of 32 —
	// costs it 2-3 points at mean runs of 2-4)
	launch_configuration_t launch_config = resolved_launch_config;
	if (dense) { launch_config.dynamic_shared_memory_size = 0; }
	stream_scratch::lease scratch(stream);
	const auto& environment = run_expansion::fine_environment::get();
	// GROUPS of segments through the long segments' one kernel (kernels/decompression/run_length_encoding.hpp, decompress_long_segments): anchors are only entry points,
	// so a workgroup may take a group of consecutive segments — about 32 768 outputs —, find the entry point of every 8192nd output of its group and decode from there.
	// Round 5, by the regret grid with the compressed input COLD (tools/regret.py, profiles/r05_regret.txt): taken for sparse columns (mean run >= 16) in whole-tile
	// segments of MORE than one finer segment and fewer than four — 12 288, 16 384, 24 576 ...: 63-66 -> 69-76 % of the HBM peak at 2^28 outputs, 37-40 -> 56-60 % at 2^26,
	// where the three-launch pre-pass used to run (its launches cost a small column more than its finer anchors gain).  NOT any more for segment lengths that are not
	// whole tiles (1000, 3000, 10 000 ...: round 4 took those this way on figures measured with one column decoded again and again — its run arrays, which the group's
	// workgroup reads twice, then come out of the Infinity Cache; from HBM the ordinary path is 5-16 points faster at 2^30 and 2^26 and equal at 2^28).
	// GIDDY_RUN_GROUP_ALL=1 in the environment: groups for every segment length (measurements).
	// (instantiations without the groups-of-runs path — other than 4-byte elements — have one build only: whatever their mean run)
	// — 8-byte elements: +2-3 points; 2-byte: from 2048 outputs on; 1-byte: never (61 -> 31)
	const bool one_kernel_build_applies = sparse || (!run_expansion::run_groups<UncompressedSize, RunLengthSize>::enabled && runs != 0 && n / runs >= 16
		&& (UncompressedSize >= 4 || (UncompressedSize == 2 && L >= 2048)));
	const bool whole_tiles_between_one_and_four_finer_segments = L % run_expansion::tile_length == 0 && L > (uint64_t) fine::sub && L < 4 * (uint64_t) fine::sub;
	if (environment.fused && one_kernel_build_applies && n > 0 && runs > 0 && (whole_tiles_between_one_and_four_finer_segments || (environment.group_all && L > 0)) && memory_access::is_aligned(run_lengths, 16)) {
		const uint64_t per_group = run_expansion::segments_per_group_for(L, UncompressedSize, fine::sub);   // (run_anchors.hpp)
		const uint64_t group_outputs = per_group * L, finer_per_group = util::div_rounding_up(group_outputs, (uint64_t) fine::sub);
		const uint64_t groups = util::div_rounding_up((uint64_t) num_segments, per_group);
		const uint64_t workgroups = groups * util::div_rounding_up(finer_per_group, (uint64_t) finer_segments_per_workgroup);
		// (up to fused_up_to_workgroups of them, as below: 2^30 outputs in segments of 24 576 at mean run 32 — 43 690 workgroups —, cold: 55 % where the three launches do 75)
		if (finer_per_group <= (uint64_t) run_expansion::fused_up_to_fines && workgroups >= environment.fused_from && per_group <= 0xFFFFFFFFull
			&& (workgroups <= run_expansion::fused_up_to_workgroups || environment.group_all)
			&& runs / groups <= fine::most_runs_per_segment_for_one_launch) {
			cuda::enqueue_launch(long_segments_kernel(), stream,
				launch_configuration_t((grid_dimension_t) std::min<uint64_t>(workgroups, 1u << 20), fine::threads),
				decompressed, run_data, run_lengths, position_anchors, intra_run_anchor_offsets,
				segment_length, num_segments, num_element_runs, uncompressed_length, (uint32_t) per_group);
			return;
		}
	}
	if (n > 0 && runs > 0 && L >= run_expansion::finer_anchors_from() && L % fine::sub == 0 && memory_access::is_aligned(run_lengths, 16) && n / runs >= environment.min_mean_run) {
		const uint64_t num_fine = util::div_rounding_up(n, (uint64_t) fine::sub);
		const uint64_t tiles = util::div_rounding_up(runs, (uint64_t) fine::runs_per_tile), blocks = util::div_rounding_up(runs, (uint64_t) fine::runs_per_block);
		// The pre-pass (kernels/decompression/run_anchors.hpp).  Three launches — sums of every 64 lengths, a scan, a search per finer anchor — read
		// the lengths twice, in parallel over the whole column: at mean runs below 8 that only pays where segments are so long that the chip is
		// not filled.  ONE launch — a workgroup per segment scanning the segment's lengths once — is a chain of memory round trips per
		// workgroup: the faster of the two where all workgroups are resident at once or nearly (at most 8192 segments: what an eighth of a
		// 2^30-output column, or a whole 2^28 one, has in segments of 32 768 or more) and runs are long (profiles/r04_run_prepass.txt:
		// 2^28 outputs, mean run 32, segments of 32 768 / 65 536: 68.7 / 70.5-71.5 % against 66.7 / 66.4 %; at 2^30 the three launches win by 1-4 points).
		const uint64_t segments = util::div_rounding_up(n, L);
		const bool one_launch = environment.one_launch && (environment.force_one_launch
			|| (sparse && L >= 4 * (uint64_t) fine::sub && segments <= fine::most_segments_for_one_launch && runs / segments <= fine::most_runs_per_segment_for_one_launch));
		// (dense columns — mean runs below 8 — from segments of four finer ones on: round 5's regret grid, cold: 32 768 / 65 536 at mean runs of 2-4: 48-50 -> 53-55 %
		// at 2^28 outputs, 19-31 -> 40 % at 2^26; round 4 had it from 131 072 on)
		const bool worth_it = environment.force || one_launch || n / runs >= 8 || L >= 4 * (uint64_t) fine::sub;
		// ... and then pre-pass and decode are ONE kernel: the workgroup that scanned a segment's lengths decodes the segment, its finer anchors never leave LDS
		// (kernels/decompression/run_length_encoding.hpp, decompress_long_segments; profiles/r04_run_prepass.txt)
		// — where there are enough segments for a workgroup each to fill the chip (2^28 outputs, mean run 32, segments of 32 768 / 65 536: 75.7 / 74.9 % against 68.9 / 71.8 %;
		// 2048 segments of 131 072: 64 % against 71 %)
		// — segments of up to 8 finer ones (a wavefront decodes two of them, one after the other: with 16 — segments of 131 072 — the separate launches win again,
		// 2^30 outputs: 75.6 % against 79.0 %; with 4 and 8: 84.2 / 82.5 % against 78.2 / 78.8 %) —
		const uint64_t workgroups_of_one_kernel = segments * util::div_rounding_up(L / fine::sub, (uint64_t) finer_segments_per_workgroup);
		// — and not so many that the launch outlasts what keeps its write front together (2^30 outputs in segments of 32 768, cold: 70 % against the three launches' 76) —
		const bool fused = environment.fused && sparse && L >= 4 * (uint64_t) fine::sub && L / fine::sub <= (uint64_t) run_expansion::fused_up_to_fines
			&& workgroups_of_one_kernel >= environment.fused_from && (workgroups_of_one_kernel <= run_expansion::fused_up_to_workgroups || environment.force_one_launch)
			&& runs / segments <= fine::most_runs_per_segment_for_one_launch;
		if (fused || (one_launch && environment.force_one_launch && environment.fused && sparse && L / fine::sub <= most_fines_per_long_segment)) {
			cuda::enqueue_launch(long_segments_kernel(), stream,
				launch_configuration_t((grid_dimension_t) std::min<uint64_t>(workgroups_of_one_kernel, 1u << 20), fine::threads),
				decompressed, run_data, run_lengths, position_anchors, intra_run_anchor_offsets,
				segment_length, num_segments, num_element_runs, uncompressed_length, (uint32_t) 1);
			return;
		}
		const size_t sums_bytes = one_launch ? 0 : (tiles + blocks) * sizeof(uint64_t);
		const size_t anchors_at = sums_bytes, offsets_at = util::round_up(anchors_at + num_fine * sizeof(run_index_type), (size_t) 16);
		if (worth_it && num_fine <= std::numeric_limits<size_type>::max() && scratch.take(offsets_at + num_fine * sizeof(run_length_type))) {
			auto* finer_anchors = reinterpret_cast<run_index_type*>(static_cast<char*>(scratch.memory()) + anchors_at);
			auto* finer_offsets = reinterpret_cast<run_length_type*>(static_cast<char*>(scratch.memory()) + offsets_at);
			if (one_launch) {
				cuda::enqueue_launch(reinterpret_cast<const void*>(fine::anchors_of_segments<RunLengthSize, RunIndexSize>), stream,
					launch_configuration_t((grid_dimension_t) std::min<uint64_t>(num_segments, 1u << 20), fine::threads),
					finer_anchors, finer_offsets, run_lengths, position_anchors, intra_run_anchor_offsets, L, (uint64_t) num_segments, runs, n);
			}
			else {
				auto* tile_starts = static_cast<uint64_t*>(scratch.memory());
				auto* block_totals = tile_starts + tiles;
				const launch_configuration_t per_block((grid_dimension_t) std::min<uint64_t>(blocks, 1u << 16), fine::threads);
				cuda::enqueue_launch(reinterpret_cast<const void*>(fine::sum_run_lengths<RunLengthSize>), stream, per_block, tile_starts, block_totals, run_lengths, runs);
				cuda::enqueue_launch(reinterpret_cast<const void*>(fine::scan_tiles<>), stream, per_block, tile_starts, static_cast<const uint64_t*>(block_totals), runs);
				cuda::enqueue_launch(reinterpret_cast<const void*>(fine::anchors_from_sums<RunLengthSize, RunIndexSize>), stream,
					launch_configuration_t((grid_dimension_t) std::min<uint64_t>(util::div_rounding_up(num_fine, (uint64_t) fine::threads), 1u << 16), fine::threads),
					finer_anchors, finer_offsets, static_cast<const uint64_t*>(tile_starts), run_lengths, position_anchors, intra_run_anchor_offsets,
					L, (uint64_t) num_segments, runs, n);
			}
			auto config = launch_config;
			const size_t wavefronts_per_block = std::max<size_t>(1, std::min<size_t>(launch_config.block_dimensions.x / warp_size, run_expansion::max_waves_per_block));
			config.grid_dimensions.x = (grid_dimension_t) std::min<uint64_t>(util::div_rounding_up(num_fine, (uint64_t) wavefronts_per_block), 1u << 20);
			cuda::enqueue_launch(decoder, stream, config,
				decompressed, run_data, run_lengths, static_cast<const run_index_type*>(finer_anchors), static_cast<const run_length_type*>(finer_offsets),
				(size_type) fine::sub, (size_type) num_fine, num_element_runs, uncompressed_length);
			return;
		}
	}
	cuda::enqueue_launch(decoder, stream, launch_config,
		decompressed, run_data, run_lengths, position_anchors, intra_run_anchor_offsets,
		segment_length, num_segments, num_element_runs, uncompressed_length);
}

/// The kernel of the reference's signature (both paths, decided per chunk: what a direct launch runs), with the two specialised builds
/// as alternates: its attributes() are limits all three accept, so a configuration resolved against them fits whichever launch() picks.
GIDDY_RLE_TEMPLATE
const device_function_t GIDDY_RLE_KERNEL::get_device_function() const
{
	return device_function_t(base_kernel(), { sparse_kernel(), dense_kernel() }, { long_segments_kernel() });
}

/// kernel_t<...>::launch as a free function: the one-line replacement for a direct launch of `decompress` ("mode C")
template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize = UncompressedIndexSize>
inline void launch(stream::id_t stream, const launch_configuration_t& launch_config,
	uint_t<UncompressedSize>* decompressed, const uint_t<UncompressedSize>* run_data, const uint_t<RunLengthSize>* run_lengths,
	const uint_t<RunIndexSize>* position_anchors, const uint_t<RunLengthSize>* intra_run_anchor_offsets,
	size_type_by_index_size<UncompressedIndexSize> segment_length, size_type_by_index_size<UncompressedIndexSize> num_segments,
	size_type_by_index_size<RunIndexSize> num_element_runs, size_type_by_index_size<UncompressedIndexSize> uncompressed_length)
{
	kernel_t<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize>::launch(stream, launch_config, decompressed, run_data, run_lengths,
		position_anchors, intra_run_anchor_offsets, segment_length, num_segments, num_element_runs, uncompressed_length);
}

#undef GIDDY_RLE_TEMPLATE
#undef GIDDY_RLE_KERNEL

} // namespace run_length_encoding
} // namespace decompression
} // namespace kernels
} // namespace cuda
