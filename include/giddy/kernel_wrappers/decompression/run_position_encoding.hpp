// Kernel wrapper for run-position decoding.  Argument-map keys and exact types follow the
// reference's src/kernel_wrappers/decompression/run_position_encoding.cu:64-105.
#pragma once
#include "giddy/kernel_wrappers/common.hpp"
#include "giddy/kernels/decompression/run_position_encoding.hpp"
#include "giddy/kernels/decompression/run_anchors.hpp"
#include "giddy/stream_scratch.hpp"

#include <cstdlib>

namespace cuda {
namespace kernels {
namespace decompression {
namespace run_position_encoding {

template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned PositionOffsetSize, bool PositionsAreRelative, unsigned RunIndexSize = UncompressedIndexSize>
class kernel_t : public cuda::registered::kernel_t {
public:
	REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(kernel_t);
	REGISTERED_KERNEL_WRAPPER_TYPED_LAUNCH(kernel_t);
	REGISTERED_KERNEL_WRAPPER_BATCHES(kernel_t);

	using run_index_type                = uint_t<RunIndexSize>;
	using uncompressed_index_type       = uint_t<UncompressedIndexSize>;
	using position_offset_type          = uint_t<PositionOffsetSize>;
	using uncompressed_type             = uint_t<UncompressedSize>;
	using index_type                    = uncompressed_index_type;
	using uncompressed_index_size_type  = size_type_by_index_size<UncompressedIndexSize>;
	using run_index_size_type           = size_type_by_index_size<RunIndexSize>;
	using position_offset_size_type     = size_type_by_index_size<PositionOffsetSize>;
	using size_type                     = uncompressed_index_size_type;
	using run_length_type               = position_offset_size_type;

	launch_configuration_t resolve_launch_configuration(
		device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
		size_t uncompressed_length, size_t uncompressed_segment_length, launch_configuration_limits_t limits) const
	{
		launch_config_resolution_params_t<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize> params(
			device_properties, kernel_function_attributes, uncompressed_length, uncompressed_segment_length);
		return cuda::kernels::resolve_launch_configuration(params, limits);
	}

	/// THE decode of a run-position encoded column on `stream`: what the type-erased enqueue_launch(), cuda::kernel::enqueue_launch(kernel, ...)
	/// with typed parameters and run_position_encoding::launch<...>() all run.  Picks the build of the decoder by the column's mean run and,
	/// for long segments, finds finer anchors first (run_anchors.hpp; scratch from stream_scratch — none to be had: the segments as they are).
	static void launch(stream::id_t stream, const launch_configuration_t& launch_config,
		uncompressed_type* decompressed, const uncompressed_type* run_data, const position_offset_type* run_start_positions,
		const run_index_type* position_anchors, position_offset_size_type uncompressed_segment_length,
		uncompressed_index_size_type num_anchors, run_index_size_type num_element_runs, uncompressed_index_size_type uncompressed_length);

	/// the argument map unpacked: sink(launch()'s parameters) — for the one launch and for a piece of a batch alike
	template <typename Sink>
	static void with_typed_parameters(const launch_configuration_t&, const arguments_type& arguments, Sink&& sink)
	{
		auto decompressed                = argument<uncompressed_type*>(arguments, "decompressed");
		auto run_data                    = argument<const uncompressed_type*>(arguments, "run_data");
		auto run_start_positions         = argument<const position_offset_type*>(arguments, "run_start_positions");
		auto position_anchors            = argument<const run_index_type*>(arguments, "position_anchors");
		auto uncompressed_segment_length = argument<position_offset_size_type>(arguments, "segment_length");
		auto num_anchors                 = argument<uncompressed_index_size_type>(arguments, "num_anchors");
		auto num_element_runs            = argument<run_index_size_type>(arguments, "num_element_runs");
		auto uncompressed_length         = argument<uncompressed_index_size_type>(arguments, "length");
		sink(decompressed, run_data, run_start_positions, position_anchors, uncompressed_segment_length, num_anchors, num_element_runs, uncompressed_length);
	}

	/// a column as one piece of a batch (giddy/batch.hpp).  hints: [0] outputs and [1] runs of all pieces — the batch's mean run picks the build of the
	/// decoder, as launch() does for one column —, [2] some piece's run arrays are not 16-byte aligned (no dense build then)
	static void add_piece(batch::table_t& table, const launch_configuration_t& piece_config,
		uncompressed_type* decompressed, const uncompressed_type* run_data, const position_offset_type* run_start_positions,
		const run_index_type* position_anchors, position_offset_size_type uncompressed_segment_length,
		uncompressed_index_size_type num_anchors, run_index_size_type num_element_runs, uncompressed_index_size_type uncompressed_length)
	{
		if (uncompressed_segment_length == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
		table.hints[0] += uncompressed_length; table.hints[1] += num_element_runs;
		if (!memory_access::is_aligned(run_start_positions, 16) || !memory_access::is_aligned(run_data, 16)) { table.hints[2] = 1; }
		table.add(piece_config, decompressed, run_data, run_start_positions, position_anchors, uncompressed_segment_length, num_anchors, num_element_runs, uncompressed_length);
	}

	/// ... and the batch's ONE launch
	static void launch_batch(stream::id_t stream, const batch::on_device_t& tables)
	{
		constexpr bool has_groups = run_expansion::run_groups<UncompressedSize, PositionOffsetSize>::enabled;
		const uint64_t outputs = tables.hints[0], runs = tables.hints[1];
		const bool sparse = has_groups && runs != 0 && outputs / runs >= 16;
		const bool dense = has_groups && runs != 0 && !sparse && tables.hints[2] == 0;
		const void* kernel = reinterpret_cast<const void*>(batch::decode_pieces<piece_body<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize, false, false>>);
		if constexpr (has_groups) {   // (the other instantiations have the one build)
			if (dense) { kernel = reinterpret_cast<const void*>(batch::decode_pieces<piece_body<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize, true, true>>); }
			else if (!sparse) { kernel = reinterpret_cast<const void*>(batch::decode_pieces<piece_body<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize, true, false>>); }
		}
		cuda::enqueue_launch(kernel, stream, launch_configuration_t(tables.num_blocks, tables.block_size, tables.dynamic_shared_memory_size), tables.pieces, tables.blocks);
	}

	/// the builds of the decoder that launch() may run with the caller's configuration
	static const void* base_kernel()   { return reinterpret_cast<const void*>(decompress<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize>); }
	static const void* sparse_kernel() { return reinterpret_cast<const void*>(decompress_sparse_runs<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize>); }
	static const void* dense_kernel()  { return reinterpret_cast<const void*>(decompress_dense_runs<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize>); }
	/// ... and, launched with a grid of its own (a workgroup per (part of an) anchored segment), the search for finer anchors and the decoder of long segments in one
	static const void* long_segments_kernel() { return reinterpret_cast<const void*>(decompress_long_segments<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize>); }
	static const void* relative_with_finer_anchors_kernel()
	{
		if constexpr (PositionsAreRelative) { return reinterpret_cast<const void*>(decompress_relative_with_finer_anchors<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, RunIndexSize>); }
		else { return nullptr; }
	}
};

#define GIDDY_RPE_TEMPLATE template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned PositionOffsetSize, bool PositionsAreRelative, unsigned RunIndexSize>
#define GIDDY_RPE_KERNEL kernel_t<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize>

GIDDY_RPE_TEMPLATE
launch_configuration_t GIDDY_RPE_KERNEL::resolve_launch_configuration(
	device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
	arguments_type extra_arguments, launch_configuration_limits_t limits) const
{
	auto uncompressed_length         = argument<size_t>(extra_arguments, "length");
	auto uncompressed_segment_length = argument<size_t>(extra_arguments, "segment_length");
	return resolve_launch_configuration(device_properties, kernel_function_attributes,
		uncompressed_length, uncompressed_segment_length, limits);
}

GIDDY_RPE_TEMPLATE
void GIDDY_RPE_KERNEL::enqueue_launch_with(stream::id_t stream, const launch_configuration_t& launch_config, const arguments_type& arguments) const
{
	with_typed_parameters(launch_config, arguments, [&](auto... parameters) { launch(stream, launch_config, parameters...); });
}

GIDDY_RPE_TEMPLATE
void GIDDY_RPE_KERNEL::launch(stream::id_t stream, const launch_configuration_t& launch_config,
	uncompressed_type* decompressed, const uncompressed_type* run_data, const position_offset_type* run_start_positions,
	const run_index_type* position_anchors, position_offset_size_type uncompressed_segment_length,
	uncompressed_index_size_type num_anchors, run_index_size_type num_element_runs, uncompressed_index_size_type uncompressed_length)
{
	// Long segments: the run covering every `fine::sub`-th output first (a binary search per finer anchor), then the decoder on
	// segments of fine::sub (kernels/decompression/run_anchors.hpp) — L = 65 536: 65 -> 8x %, L = 2^20: 26 -> 8x % of the HBM peak.
	namespace fine = run_expansion::fine;
	const uint64_t n = uncompressed_length, L = uncompressed_segment_length;
	// mean runs of 16 or more: the kernel without the groups-of-runs path (kernels/decompression/run_position_encoding.hpp)
	const bool sparse = run_expansion::run_groups<UncompressedSize, PositionOffsetSize>::enabled && num_element_runs != 0 && n / num_element_runs >= 16;
	// ... below 16, run arrays 16-byte aligned: the kernel with ONLY that path
	const bool dense = run_expansion::run_groups<UncompressedSize, PositionOffsetSize>::enabled && num_element_runs != 0 && !sparse
		&& memory_access::is_aligned(run_start_positions, 16) && memory_access::is_aligned(run_data, 16);
	const void* const decoder = sparse ? sparse_kernel() : dense ? dense_kernel() : base_kernel();
	stream_scratch::lease scratch(stream);
	// Groups of segments through the long segments' one kernel: round 4 took segment lengths that are not whole tiles this way, on figures measured with the run
	// arrays in the Infinity Cache; with them COLD (round 5, tools/regret.py) the ordinary path is the faster one — relative positions in segments of 8200: 58 -> 67-70 % of
	// the HBM peak, absolute: 63 -> 69 — so only GIDDY_RUN_GROUP_ALL=1 (measurements) still does
	if (run_expansion::fine_environment::get().fused && run_expansion::fine_environment::get().group_all && sparse && UncompressedSize >= 4 && n > 0 && num_element_runs > 0 && (PositionsAreRelative ? L >= 256 : L > 0)) {   // (a piece of 4096 outputs: within 17 segments of relative positions)
		const uint64_t per_group = run_expansion::segments_per_group_for(L, UncompressedSize, fine::sub);
		const uint64_t finer_per_group = util::div_rounding_up(per_group * L, (uint64_t) fine::sub);
		const uint64_t groups = util::div_rounding_up((uint64_t) num_anchors, per_group);
		const uint64_t workgroups = groups * util::div_rounding_up(finer_per_group, (uint64_t) run_length_encoding::finer_segments_per_workgroup);
		if (finer_per_group <= (uint64_t) run_expansion::fused_up_to_fines && workgroups >= run_expansion::fine_environment::get().fused_from && per_group <= 0xFFFFFFFFull) {
			cuda::enqueue_launch(long_segments_kernel(), stream,
				launch_configuration_t((grid_dimension_t) std::min<uint64_t>(workgroups, 1u << 20), fine::threads),
				decompressed, run_data, run_start_positions, position_anchors, L, (run_index_size_type) num_anchors, num_element_runs, uncompressed_length, (uint32_t) per_group);
			return;
		}
	}
	// (relative positions of DENSE columns — mean runs below 16 — in segments of fewer than four finer ones keep their segments: 16 384 / 24 576 at mean runs of 2-4: 47-51 -> 54-57 %, cold;
	// and in longer ones where there are segments enough to fill the chip with a wavefront each — 16 384: 2^30 outputs in segments of 32 768 / 65 536 at mean runs of 2-4, cold:
	// 50-56 -> 56-60 %; 2^28 outputs, 8192 / 4096 segments: the finer anchors win by 1-4 points)
	// (... at mean runs of 2-4: at 8 the finer anchors win there, 62 % against 56-59)
	const uint64_t mean_run = n / std::max<uint64_t>(num_element_runs, 1);
	const bool dense_relative_as_it_is = PositionsAreRelative && ((mean_run < 16 && L < 4 * (uint64_t) fine::sub) || (mean_run < 6 && (uint64_t) num_anchors >= 16384))
		&& !run_expansion::fine_environment::get().force;
	if (n > 0 && num_element_runs > 0 && L >= run_expansion::finer_anchors_from() && L % fine::sub == 0 && !dense_relative_as_it_is) {
		const uint64_t num_fine = util::div_rounding_up(n, (uint64_t) fine::sub);
		// ... in ONE kernel where segments are of 4 finer ones and there are enough of them for 2048 workgroups (kernels/decompression/run_position_encoding.hpp,
		// decompress_long_segments).  The search is a cheap launch of its own, unlike the run lengths' scan: 2^28 outputs in segments of 32 768: 76.9 % against 74-75 %
		// (2^30: 82-85 against 79-80), but of 65 536 — a wavefront decodes two finer segments one after the other: 70-72 % against 74-75 % (profiles/r04_run_prepass.txt)
		const uint64_t workgroups_of_one_kernel = (uint64_t) num_anchors * util::div_rounding_up(L / fine::sub, (uint64_t) run_length_encoding::finer_segments_per_workgroup);
		if (run_expansion::fine_environment::get().fused && sparse && L >= 4 * (uint64_t) fine::sub && L / fine::sub <= (uint64_t) run_expansion::positions_fused_up_to_fines
			&& workgroups_of_one_kernel >= run_expansion::fine_environment::get().fused_from
			// — and, as for run lengths, not more workgroups than keep a launch's write front together (8192: run_anchors.hpp): round 5, the regret grid at 2^29 and 2^30 outputs with the input COLD
			// (profiles/r05_regret.txt): segments of 32 768 at mean runs of 32-64, relative 57-62 % against the search as a launch of its own 69-70 %, absolute 62-65 against 75-76
			// (round 4's 82-85 % were measured with one column decoded again and again)
			&& (workgroups_of_one_kernel <= run_expansion::positions_fused_up_to_workgroups || run_expansion::fine_environment::get().force_one_launch)) {
			cuda::enqueue_launch(long_segments_kernel(), stream,
				launch_configuration_t((grid_dimension_t) std::min<uint64_t>(workgroups_of_one_kernel, 1u << 20), fine::threads),
				decompressed, run_data, run_start_positions, position_anchors, L, (run_index_size_type) num_anchors, num_element_runs, uncompressed_length, (uint32_t) 1);
			return;
		}
		if (num_fine <= std::numeric_limits<run_index_size_type>::max() && scratch.take(num_fine * sizeof(run_index_type))) {
			auto* finer_anchors = static_cast<run_index_type*>(scratch.memory());
			cuda::enqueue_launch(reinterpret_cast<const void*>(fine::anchors_by_search<PositionOffsetSize, PositionsAreRelative, RunIndexSize>), stream,
				launch_configuration_t((grid_dimension_t) std::min<uint64_t>(util::div_rounding_up(num_fine, (uint64_t) fine::threads), 1u << 16), fine::threads),
				finer_anchors, run_start_positions, position_anchors, L, (uint64_t) num_anchors, (uint64_t) num_element_runs, n);
			auto config = launch_config;
			const size_t wavefronts_per_block = std::max<size_t>(1, std::min<size_t>(launch_config.block_dimensions.x / warp_size, run_expansion::max_waves_per_block));
			config.grid_dimensions.x = (grid_dimension_t) std::min<uint64_t>(util::div_rounding_up(num_fine, (uint64_t) wavefronts_per_block), 1u << 20);
			if (PositionsAreRelative) {
				cuda::enqueue_launch(relative_with_finer_anchors_kernel(), stream, config,
					decompressed, run_data, run_start_positions, static_cast<const run_index_type*>(finer_anchors),
					(uint64_t) fine::sub, L, (run_index_size_type) num_fine, num_element_runs, uncompressed_length);
			}
			else {
				cuda::enqueue_launch(decoder, stream, config,
					decompressed, run_data, run_start_positions, static_cast<const run_index_type*>(finer_anchors),
					(position_offset_size_type) fine::sub, (run_index_size_type) num_fine, num_element_runs, uncompressed_length);
			}
			return;
		}
	}
	cuda::enqueue_launch(decoder, stream, launch_config,
		decompressed, run_data, run_start_positions, position_anchors,
		uncompressed_segment_length, (run_index_size_type) num_anchors, num_element_runs, uncompressed_length);
}

/// The kernel of the reference's signature (both paths, decided per chunk: what a direct launch runs), with the specialised builds as
/// alternates: its attributes() are limits all of them accept, so a configuration resolved against them fits whichever launch() picks.
GIDDY_RPE_TEMPLATE
const device_function_t GIDDY_RPE_KERNEL::get_device_function() const
{
	return device_function_t(base_kernel(), { sparse_kernel(), dense_kernel(), relative_with_finer_anchors_kernel() }, { long_segments_kernel() });
}

/// kernel_t<...>::launch as a free function: the one-line replacement for a direct launch of `decompress` ("mode C")
template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned PositionOffsetSize, bool PositionsAreRelative, unsigned RunIndexSize = UncompressedIndexSize>
inline void launch(stream::id_t stream, const launch_configuration_t& launch_config,
	uint_t<UncompressedSize>* decompressed, const uint_t<UncompressedSize>* run_data, const uint_t<PositionOffsetSize>* run_start_positions,
	const uint_t<RunIndexSize>* position_anchors, size_type_by_index_size<PositionOffsetSize> segment_length,
	size_type_by_index_size<UncompressedIndexSize> num_anchors, size_type_by_index_size<RunIndexSize> num_element_runs,
	size_type_by_index_size<UncompressedIndexSize> uncompressed_length)
{
	kernel_t<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize>::launch(stream, launch_config,
		decompressed, run_data, run_start_positions, position_anchors, segment_length, num_anchors, num_element_runs, uncompressed_length);
}

#undef GIDDY_RPE_TEMPLATE
#undef GIDDY_RPE_KERNEL

} // namespace run_position_encoding
} // namespace decompression
} // namespace kernels
} // namespace cuda
