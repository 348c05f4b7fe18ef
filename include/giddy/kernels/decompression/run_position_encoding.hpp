// Run-position decoding for CDNA4:   out[p] = run_data[r(p)],  r(p) = the last run whose start is <= p.
// Run starts are stored directly — absolute, or relative to the start of the anchored segment the run
// begins in (PositionsAreRelative) — and position_anchors[s] is the run covering output s*L.
//
// Same kernel template name, parameter list and element semantics as the reference's
// src/kernels/decompression/run_position_encoding.cuh:327-359 (per-segment parameters :255-308), with
// one deliberate difference: a run that starts exactly on a segment boundary belongs to the NEXT
// segment only.  The reference admits it into the previous segment when positions are relative
// (it compares a relative position with an absolute boundary, :296-297) and then mis-decodes that
// segment's last run; here the test is "0 < relative start < L".
//
// Execution (see run_expansion.hpp): one WAVEFRONT per anchored segment at grid stride, tiles of
// <= 4096 outputs; no scan is needed — start positions are marked straight into the tile bitmap,
// 4 candidates per lane per batch, until a batch reaches past the tile; rank; expand.  For segments
// longer than one tile the last run starting at or before the tile's end is carried forward.
//
// HBM bytes per output: element + (element + position)/mean_run + anchor/L  (4.19 B relative /
// 4.25 B absolute at 32-bit values, mean run 32, L = 4096 — SURVEY.md §8d, config C4).
#pragma once
#include <hip/hip_runtime.h>

#include "giddy/common.hpp"
#include "giddy/kernels/decompression/run_expansion.hpp"
#include "giddy/kernels/decompression/run_anchors.hpp"
#include "giddy/kernels/decompression/run_length_encoding.hpp"   // (chunks_to_take, the long-segment kernels' constants)
#include "giddy/kernels/launch_config_resolution_params.hpp"
#include "giddy/util/math.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace run_position_encoding {

namespace detail {

/// segments a wavefront takes per trip around its outer loop: as many short ones as fill one tile — with relative
/// positions at most 32, the reach of the search over the chunk's anchors
__host__ __device__ inline uint64_t segments_per_wave_chunk(uint64_t segment_length, bool positions_are_relative, unsigned element_size = 4)
{
	if (segment_length >= run_expansion::tile_length) {   // (1-byte columns: run_length_encoding.hpp)
		const uint64_t fit = element_size == 1 && !positions_are_relative ? (uint64_t) run_expansion::narrow_chunk_bytes / segment_length   // (relative positions: a search over the chunk's anchors per candidate costs more than the trip saves)
			: 1;
		return fit > 1 ? fit : 1;
	}
	uint64_t fit = run_expansion::tile_length / segment_length;
	if (fit < run_expansion::segments_per_trip_at_least) { fit = run_expansion::segments_per_trip_at_least; }   // (segments of more than half a tile go two at a time: two tiles with a carry beat one half-empty tile)
	fit = (positions_are_relative && fit > 32) ? 32 : fit;
	return run_expansion::whole_64_byte_blocks(fit, segment_length, element_size);
}

} // namespace detail

namespace detail {

/// The kernel's body.  `position_period`: what RELATIVE positions are relative to — the start of the last multiple of it at or
/// before the run; the anchoring period, except when the wrapper has put finer anchors into long segments (run_anchors.hpp: the
/// anchors then come every `anchoring_period`, a divisor of it and at least a tile long, while the positions still count from
/// the starts of the segments the column was compressed in).
template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned PositionOffsetSize, bool PositionsAreRelative, unsigned RunIndexSize, bool WithGroups = true, bool GroupsAlways = false, typename Grid = launch_grid::whole>
__device__ __forceinline__ void decompress(
	uint_t<UncompressedSize>*          __restrict__  decompressed,
	const uint_t<UncompressedSize>*    __restrict__  run_data,
	const uint_t<PositionOffsetSize>*  __restrict__  run_start_positions,
	const uint_t<RunIndexSize>*        __restrict__  position_anchors,
	uint64_t                                         anchoring_period,
	uint64_t                                         position_period,
	size_type_by_index_size<RunIndexSize>            num_anchors,
	size_type_by_index_size<RunIndexSize>            num_element_runs,
	size_type_by_index_size<UncompressedIndexSize>   uncompressed_length,
	const run_length_encoding::detail::chunks_to_take mine = run_length_encoding::detail::chunks_to_take(),   // (decompress_long_segments: one workgroup's chunks, its anchors in LDS)
	Grid                                             grid = Grid())                                            // (launch_grid.hpp: the launch's own, or a batch piece's part of it)
{
	static_assert(PositionsAreRelative || PositionOffsetSize >= UncompressedIndexSize,
		"If run positions are in absolute values, their type must be able to cover the entire potential range of data");
	using namespace run_expansion;
	__shared__ tile_entry tables[max_waves_per_block][tile_words];
	using staging = value_staging<UncompressedSize, PositionOffsetSize>;
	__shared__ alignas(16) uint_t<UncompressedSize> staged_values[max_waves_per_block][staging::allocated];
	uint_t<UncompressedSize>* staged = staged_values[wave64::index_in_block() % max_waves_per_block];   // (idle wavefronts never use it)
	// chunks of many runs: positions and values in GROUPS of 512 runs through LDS (run_expansion.hpp: run_groups; run_length_encoding.hpp)
	using groups = run_groups<UncompressedSize, PositionOffsetSize, WithGroups>;
	__shared__ alignas(16) uint_t<PositionOffsetSize> group_positions_storage[groups::enabled ? max_waves_per_block : 1][groups::enabled ? groups::allocated : 1];
	uint_t<PositionOffsetSize>* group_positions = group_positions_storage[groups::enabled ? wave64::index_in_block() % max_waves_per_block : 0];

	// LDS is provisioned for max_waves_per_block wavefronts (the launch parameters cap the workgroup at 256
	// threads); in a larger workgroup the extra wavefronts sit idle — there is no barrier they could miss
	const unsigned working_waves = wave64::per_block() < max_waves_per_block ? wave64::per_block() : max_waves_per_block;
	if (wave64::index_in_block() >= working_waves) { return; }
	const unsigned lane = wave64::lane_id();
	tile_entry* table = tables[wave64::index_in_block()];
	const uint64_t wave = (uint64_t) grid.block() * working_waves + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) grid.blocks() * working_waves;
	const uint64_t L = anchoring_period, n = uncompressed_length, num_runs = num_element_runs;

	// Short segments are taken a tile's worth at a time — the runs of consecutive segments are consecutive, so k segments
	// entered at the first one's anchor decode like one segment of k * L outputs.  Relative positions restart in every
	// segment: a run's segment within the chunk is found by searching the chunk's anchors (one per lane).
	const uint64_t segments_per_chunk = detail::segments_per_wave_chunk(L, PositionsAreRelative, UncompressedSize);
	const uint64_t num_chunks = ((uint64_t) num_anchors + segments_per_chunk - 1) / segments_per_chunk;
	// (4- and 8-byte elements only: narrow columns lose 2-4 points to it even with the output on a page boundary — 41-repetition A/B)
	const uint64_t my_first_chunk = UncompressedSize >= 4 ? first_chunk_by_address(decompressed, segments_per_chunk * L * UncompressedSize, working_waves, wave, num_waves) : wave;
	for (uint64_t c = mine.given ? mine.first : my_first_chunk; c < (mine.given ? mine.limit : num_chunks); c += mine.given ? mine.stride : num_waves) {
		const uint64_t s = c * segments_per_chunk;
		const uint64_t segment_start = mine.given ? mine.outputs_from + (s - mine.anchors_from) * L : s * L;
		if (segment_start >= n) { break; }
		// where the first segment after this chunk starts (a GROUP of segments' last finer segment — decompress_long_segments — ends with the group)
		const uint64_t nominal_end = segment_start + segments_per_chunk * L < mine.outputs_until ? segment_start + segments_per_chunk * L : mine.outputs_until;
		const uint64_t segment_end = nominal_end < n ? nominal_end : n;
		// (relative positions in a piece of a GROUP of the column's segments — decompress_long_segments —: the piece may lie across several of them; which one a run
		// is in is found as for a batch of short segments below, over the COLUMN's anchors of the segments the piece touches)
		const bool regrouped = PositionsAreRelative && mine.given && mine.segment_anchors != nullptr;   // (wave-uniform)
		const uint64_t relative_to = regrouped ? mine.segment_period : position_period;
		const uint64_t positions_origin = relative_to == L ? segment_start : segment_start / relative_to * relative_to;
		const uint64_t segment_stride = regrouped ? relative_to : L;   // from one segment's start to the next one's, for the positions
		const unsigned segments_here = regrouped ? (unsigned) ((segment_end - 1) / relative_to - segment_start / relative_to + 1) : (unsigned) segments_per_chunk;
		// (both anchors asked for at once — the second at an index that always exists — and ONE wait: a conditional second load is a second round trip, and with the
		// compressed input cold a round trip is 2 microseconds of a wavefront's 17; round 5)
		const bool a_chunk_follows = s + segments_per_chunk < num_anchors;
		const uint64_t this_anchor = position_anchors[s - mine.anchors_from], next_anchor = position_anchors[(a_chunk_follows ? s + segments_per_chunk : s) - mine.anchors_from];
		uint64_t first_run = this_anchor;
		// candidates for runs starting inside this chunk: first_run+1 .. last_candidate (inclusive)
		const uint64_t last_candidate = a_chunk_follows ? next_anchor : num_runs - 1;
		// lane j: the anchor of the chunk's j-th segment (past the chunk / the column: no run index reaches it)
		const uint64_t chunk_anchor = regrouped
			? (lane < segments_here ? (uint64_t) static_cast<const uint_t<RunIndexSize>*>(mine.segment_anchors)[segment_start / relative_to + lane] : ~(uint64_t) 0)
			: (PositionsAreRelative && segments_per_chunk > 1 && lane < segments_per_chunk && s + lane < num_anchors)
				? (uint64_t) position_anchors[s + lane - mine.anchors_from] : ~(uint64_t) 0;

		// (... as 32-bit distances from the chunk's first anchor — a chunk's anchors lie within a chunk's runs of each other —: the search below moves them between lanes with
		// one ds_bpermute a step, not two, and takes as many steps as the chunk has segments to tell apart: two for four segments of 1024, not five)
		const uint64_t first_anchor_of_chunk = wave64::read_first_lane(chunk_anchor);
		const uint32_t chunk_anchor_distance = chunk_anchor == ~(uint64_t) 0 || chunk_anchor - first_anchor_of_chunk > 0xFFFFFFF0ull ? 0xFFFFFFFFu : (uint32_t) (chunk_anchor - first_anchor_of_chunk);
		const unsigned first_search_step = segments_here > 1 ? 1u << (31 - __builtin_clz(segments_here - 1)) : 0u;   // (wave-uniform)

		// groups of runs through LDS when the chunk has more runs than one batch and the arrays allow 16-byte loads
		const uint64_t run_limit = last_candidate + 1;   // (exclusive: the runs this chunk may look at)
		// (`GroupsAlways`: the wrapper's launch for dense columns, arrays 16-byte aligned: every chunk by groups — the other path is compiled out)
		const bool use_groups = (GroupsAlways && groups::enabled) || (groups::enabled && last_candidate - first_run > runs_per_batch
			&& memory_access::is_aligned(run_start_positions, 16) && memory_access::is_aligned(run_data, 16));
		uint64_t group_base = 0, group_end = 0, next_group_base = ~(uint64_t) 0;   // runs [group_base, group_end) are in LDS; the group on its way
		typename groups::in_flight coming = { };
		auto request_group = [&](uint64_t base) { next_group_base = base; groups::request(coming, run_start_positions, run_data, base, num_runs, lane); };
		auto take_group = [&]() {
			groups::into_lds(coming, group_positions, staged, lane);
			group_base = next_group_base;
			group_end = group_base + groups::group_runs < run_limit ? group_base + groups::group_runs : run_limit;
			next_group_base = ~(uint64_t) 0;
			if (group_end < run_limit) { request_group((group_end - groups::overlap) / 8 * 8); }
		};
		if (use_groups) { request_group(first_run / 8 * 8); take_group(); }

		// the NEXT tile's first batch of run positions (and values) is requested before this tile's expansion stores anything
		// (run_length_encoding.hpp: loads issued after a tile's stores wait for those stores as well)
		uint_t<PositionOffsetSize> prefetched_positions[runs_per_lane];
		uint_t<UncompressedSize> prefetched_values[runs_per_lane] = { }, prefetched_first_value = 0;
		bool prefetched = false, values_prefetched = false;
		uint32_t tile_len = 0;
		for (uint64_t tile_start = segment_start; tile_start < segment_end && first_run <= last_candidate; tile_start += tile_len) {
			tile_len = (uint32_t) ((segment_end - tile_start < tile_length) ? segment_end - tile_start : tile_length);
			bool need_carry = tile_start + tile_len < segment_end;
			bool dense = false, values_staged = false;
			if (use_groups && next_group_base <= first_run) { take_group(); }   // (the tile before was cut at the old group's end)
			clear(table, lane);
			if (!staging::with_descriptors && prefetched && values_prefetched && lane == 0) { staged[0] = prefetched_first_value; }

			// Candidates are addressed RELATIVE to the tile's first run from here on (candidate k is run first_run + 1 + k; 32 bits: a
			// tile holds at most tile_length of them) and a batch's loads are unconditional — a lane past the last candidate re-reads
			// that one and discards it — so that all of them are in flight together: no branch, no wait between one load and the next.
			// (groups: a tile sees the candidates that are in LDS, and one batch of them at most while there are runs beyond what it sees — it
			// is then cut short at the last start it saw, as a dense tile is)
			const uint64_t last_visible = !use_groups ? last_candidate
				: (group_end <= last_candidate && first_run + runs_per_batch < group_end - 1 ? first_run + runs_per_batch : (group_end - 1 < last_candidate ? group_end - 1 : last_candidate));
			const bool more_beyond = last_visible < last_candidate;
			const uint32_t num_candidates = last_visible - first_run > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t) (last_visible - first_run);
			const uint_t<PositionOffsetSize>* staged_positions_from_here = group_positions + (first_run - group_base);
			const uint_t<PositionOffsetSize>* __restrict__ positions_from_here = run_start_positions + first_run;
			const uint_t<UncompressedSize>* __restrict__ values_from_here = run_data + first_run;
			// a start is of interest up to the tile's end; one at/after the chunk's nominal end opens the NEXT chunk
			const uint64_t reach = nominal_end - tile_start < tile_length + 1 ? nominal_end - tile_start : tile_length + 1;
			uint32_t batch_base = 0;
			uint32_t last_at_or_before_end = 0;   // 1 + (index relative to first_run) of the last run starting <= the tile's end
			bool reached_past_tile;
			do {
				uint_t<PositionOffsetSize> positions[runs_per_lane]; uint_t<UncompressedSize> values[runs_per_lane] = { };
				uint32_t offsets[runs_per_lane];   // of a candidate's start from the tile's; ~0: no such candidate, or it starts past the tile / the chunk
				const uint32_t lane_base = batch_base + lane * runs_per_lane;
				const bool from_prefetch = prefetched && batch_base == 0;   // (wave-uniform)
				const bool stage_prefetched = !staging::with_descriptors && from_prefetch && values_prefetched;   // a dense tile's successor
				if (from_prefetch) {
					#pragma unroll
					for (unsigned j = 0; j < runs_per_lane; j++) { positions[j] = prefetched_positions[j]; values[j] = prefetched_values[j]; }
				}
				else if (use_groups) {
					#pragma unroll
					for (unsigned j = 0; j < runs_per_lane; j++) {
						const uint32_t r = lane_base + j < num_candidates ? lane_base + j + 1 : num_candidates;   // (relative to first_run)
						positions[j] = staged_positions_from_here[r];   // (the values are in LDS already, at the same offsets)
					}
				}
				else {
					#pragma unroll
					for (unsigned j = 0; j < runs_per_lane; j++) {
						const uint32_t r = lane_base + j < num_candidates ? lane_base + j + 1 : num_candidates;   // (relative to first_run)
						positions[j] = load_descriptor(positions_from_here + r);
						if (staging::with_descriptors) { values[j] = values_from_here[r]; }   // the runs' values travel with their positions
					}
				}
				// the tile's first run's value — the run that reaches into the tile is no candidate —: asked for here, behind the batch's loads, not ahead of them (there it
				// was a round trip of its own before the batch's: the other half of what run positions lost against run lengths with the input cold)
				if (batch_base == 0 && !use_groups && staging::with_descriptors && lane == 0) { staged[0] = prefetched ? prefetched_first_value : run_data[first_run]; }
				#pragma unroll
				for (unsigned j = 0; j < runs_per_lane; j++) {
					const uint32_t k = lane_base + j;
					const bool have = k < num_candidates;
					uint64_t start = positions[j];
					if (PositionsAreRelative) {
						// Relative positions: the run lies in the chunk's segment i = #{j >= 1 : anchor_j < r} — a binary search over
						// the anchors held one per lane
						unsigned segment_in_chunk = 0;
						if (segments_here > 1) {   // wave-uniform
							const uint64_t r = first_run + 1 + k - first_anchor_of_chunk;
							const uint32_t distance = r > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t) r;
							auto search_step = [&](unsigned step) {
								const unsigned probe = segment_in_chunk + step;
								const uint32_t anchor = (uint32_t) __shfl((int) chunk_anchor_distance, (int) (probe < 64 ? probe : 63), 64);
								if (probe < segments_here && anchor < distance) { segment_in_chunk = probe; }
							};
							if (first_search_step == 16) {   // (17 ... 32 segments — of 128 outputs and fewer —: all five steps, unrolled: as a loop they cost such columns 1-3 points)
								#pragma unroll
								for (unsigned step = 16; step >= 1; step >>= 1) { search_step(step); }
							}
							else { for (unsigned step = first_search_step; step >= 1; step >>= 1) { search_step(step); } }   // (a wave-uniform number of turns)
						}
						// ... and at relative position 0 it starts the NEXT segment
						start = positions_origin + (uint64_t) segment_in_chunk * segment_stride + (positions[j] != 0 ? (uint64_t) positions[j] : relative_to);
					}
					const uint64_t from_tile_start = start - tile_start;
					offsets[j] = (have && from_tile_start < reach) ? (uint32_t) from_tile_start : ~0u;
					// (staged for the expansion; runs past the tile's last are harmless)
					if (!use_groups && (staging::with_descriptors || stage_prefetched) && have && k + 1 < staging::capacity) { staged[k + 1] = values[j]; }
				}
				values_staged = values_staged || stage_prefetched;
				if (batch_base == 0) {
					// a dense tile (run_expansion.hpp): when every run of the first batch starts inside the tile, the tile ends where
					// the batch's last run starts (on a whole 128-byte piece of output)
					bool outside = false;
					uint32_t furthest_inside = 0;
					#pragma unroll
					for (unsigned j = 0; j < runs_per_lane; j++) {
						const bool visible = lane_base + j < num_candidates;
						// (groups: the candidates the tile cannot see are neither inside nor outside — if all it sees start inside and there are more, it is cut)
						outside = outside || (offsets[j] >= tile_len && (!more_beyond || visible));
						furthest_inside = (visible && offsets[j] < tile_len && offsets[j] > furthest_inside) ? offsets[j] : furthest_inside;
					}
					const uint32_t last_start = more_beyond ? wave64::read_last_lane(wave64::inclusive_max(furthest_inside)) : wave64::read_last_lane(offsets[runs_per_lane - 1]);
					if (wave64::ballot(outside) == 0 && last_start >= staging::dense_granule && last_start < tile_len) {   // (positions that do not increase: no empty tile)
						tile_len = last_start / staging::dense_granule * staging::dense_granule;
						need_carry = true; dense = true;
					}
				}
				bool past = false;
				#pragma unroll
				for (unsigned j = 0; j < runs_per_lane; j++) {
					const bool at_or_before_end = offsets[j] <= tile_len;
					past = past || !at_or_before_end;
					if (offsets[j] - 1u < tile_len - 1u) { mark(table, offsets[j]); }   // 0 < offset < tile_len
					last_at_or_before_end = at_or_before_end ? lane_base + j + 2 : last_at_or_before_end;   // increases with j
				}
				reached_past_tile = wave64::ballot(past) != 0;
				batch_base += runs_per_batch;
			} while (!reached_past_tile);

			// where the next tile enters the runs — and its first batch, on its way before this tile stores
			uint64_t next_first_run = first_run;
			prefetched = false;
			if (need_carry) {
				uint32_t furthest = wave64::read_last_lane(wave64::inclusive_max(last_at_or_before_end));
				if (furthest != 0) { next_first_run = first_run + furthest - 1; }
				if (UncompressedSize >= prefetch_from_element_size && !use_groups) {   // (1-byte columns: 16 outputs per lane, the registers are worth more)
					const uint32_t candidates_there = last_candidate - next_first_run > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t) (last_candidate - next_first_run);
					uint32_t r[runs_per_lane];
					#pragma unroll
					for (unsigned j = 0; j < runs_per_lane; j++) {
						r[j] = lane * runs_per_lane + j < candidates_there ? lane * runs_per_lane + j + 1 : candidates_there;   // (relative to next_first_run)
						prefetched_positions[j] = load_descriptor(run_start_positions + next_first_run + r[j]);
					}
					prefetched = true; values_prefetched = staging::with_descriptors || dense;   // (after a dense tile, most likely another)
					if (values_prefetched) {
						#pragma unroll
						for (unsigned j = 0; j < runs_per_lane; j++) { prefetched_values[j] = run_data[next_first_run + r[j]]; }
						prefetched_first_value = run_data[next_first_run];
					}
				}
			}

			const uint32_t marks = rank(table, lane);
			constexpr int staged_stores = staged_store_policy<UncompressedSize, PositionOffsetSize, PositionsAreRelative>();
			if (use_groups) {   // the tile's values are in LDS, at the group's offsets
				expand<UncompressedSize, true, true, staged_stores>(decompressed + tile_start, tile_len, table, run_data + first_run, lane, staged + (first_run - group_base), marks + 1);
			}
			else if (staging::dense_only && !dense) {
				expand<UncompressedSize, false>(decompressed + tile_start, tile_len, table, run_data + first_run, lane);
			}
			else if (!staging::with_descriptors && values_staged) {   // a dense tile whose values came with the prefetched positions
				expand<UncompressedSize, true, true, staged_stores>(decompressed + tile_start, tile_len, table, run_data + first_run, lane, staged, marks + 1);
			}
			else {
				expand<UncompressedSize, true, staging::with_descriptors, staged_stores>(decompressed + tile_start, tile_len, table, run_data + first_run, lane, staged, marks + 1);
			}
			first_run = next_first_run;
		}
	}
}

} // namespace detail

template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned PositionOffsetSize, bool PositionsAreRelative, unsigned RunIndexSize = UncompressedIndexSize>
__global__ void decompress(
	uint_t<UncompressedSize>*          __restrict__  decompressed,
	const uint_t<UncompressedSize>*    __restrict__  run_data,
	const uint_t<PositionOffsetSize>*  __restrict__  run_start_positions,
	const uint_t<RunIndexSize>*        __restrict__  position_anchors,
	size_type_by_index_size<PositionOffsetSize>      anchoring_period,
	size_type_by_index_size<RunIndexSize>            num_anchors,
	size_type_by_index_size<RunIndexSize>            num_element_runs,
	size_type_by_index_size<UncompressedIndexSize>   uncompressed_length)
{
	detail::decompress<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize>(
		decompressed, run_data, run_start_positions, position_anchors, anchoring_period, anchoring_period, num_anchors, num_element_runs, uncompressed_length);
}

/// The same decoder with ONLY the groups-of-runs path: the wrapper's launch for columns of mean runs below 16 whose run arrays are 16-byte aligned
template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned PositionOffsetSize, bool PositionsAreRelative, unsigned RunIndexSize = UncompressedIndexSize>
__global__ void decompress_dense_runs(
	uint_t<UncompressedSize>*          __restrict__  decompressed,
	const uint_t<UncompressedSize>*    __restrict__  run_data,
	const uint_t<PositionOffsetSize>*  __restrict__  run_start_positions,
	const uint_t<RunIndexSize>*        __restrict__  position_anchors,
	size_type_by_index_size<PositionOffsetSize>      anchoring_period,
	size_type_by_index_size<RunIndexSize>            num_anchors,
	size_type_by_index_size<RunIndexSize>            num_element_runs,
	size_type_by_index_size<UncompressedIndexSize>   uncompressed_length)
{
	detail::decompress<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize, true, true>(
		decompressed, run_data, run_start_positions, position_anchors, anchoring_period, anchoring_period, num_anchors, num_element_runs, uncompressed_length);
}

/// The same decoder without the path for chunks of many runs (groups of runs through LDS: run_expansion.hpp) — 15-17 registers and 4-8 KiB of
/// LDS lighter, and its waits are not the stricter of two paths': at mean runs of 32 with the compressed input NOT in the Infinity Cache it is
/// 9-12 points faster (16-bit relative positions 62 -> 74 %, 32-bit absolute 67 -> 76 %), with the input cached 3 points slower or 6 faster; the
/// wrapper launches it for columns whose mean run is 16 or more, where no chunk has enough runs for a group anyway.
template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned PositionOffsetSize, bool PositionsAreRelative, unsigned RunIndexSize = UncompressedIndexSize>
__global__ void decompress_sparse_runs(
	uint_t<UncompressedSize>*          __restrict__  decompressed,
	const uint_t<UncompressedSize>*    __restrict__  run_data,
	const uint_t<PositionOffsetSize>*  __restrict__  run_start_positions,
	const uint_t<RunIndexSize>*        __restrict__  position_anchors,
	size_type_by_index_size<PositionOffsetSize>      anchoring_period,
	size_type_by_index_size<RunIndexSize>            num_anchors,
	size_type_by_index_size<RunIndexSize>            num_element_runs,
	size_type_by_index_size<UncompressedIndexSize>   uncompressed_length)
{
	detail::decompress<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize, false>(
		decompressed, run_data, run_start_positions, position_anchors, anchoring_period, anchoring_period, num_anchors, num_element_runs, uncompressed_length);
}

/// A piece of a batch launch (giddy/batch.hpp; launched as batch::decode_pieces<piece_body<...>>): `decompress`'s parameters out of the piece's record, the body
/// — with or without the groups-of-runs path, as the three kernels above; the wrapper picks by the batch's mean run — on the piece's part of the grid
template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned PositionOffsetSize, bool PositionsAreRelative, unsigned RunIndexSize, bool WithGroups, bool GroupsAlways>
struct piece_body {
	enum : unsigned { max_threads_per_block = 1024 };
	static __device__ __forceinline__ void run(const batch::piece_view piece, const launch_grid::part grid)
	{
		const uint64_t anchoring_period = piece.get<size_type_by_index_size<PositionOffsetSize>>(4);
		detail::decompress<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize, WithGroups, GroupsAlways>(
			piece.get<uint_t<UncompressedSize>*>(0), piece.get<const uint_t<UncompressedSize>*>(1), piece.get<const uint_t<PositionOffsetSize>*>(2),
			piece.get<const uint_t<RunIndexSize>*>(3), anchoring_period, anchoring_period,
			piece.get<size_type_by_index_size<RunIndexSize>>(5), piece.get<size_type_by_index_size<RunIndexSize>>(6), piece.get<size_type_by_index_size<UncompressedIndexSize>>(7),
			run_length_encoding::detail::chunks_to_take(), grid);
	}
};

/// The same decoder over anchors finer than the segments the relative positions count from (the wrapper's launch for long segments)
template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned PositionOffsetSize, unsigned RunIndexSize = UncompressedIndexSize>
__global__ void decompress_relative_with_finer_anchors(
	uint_t<UncompressedSize>*          __restrict__  decompressed,
	const uint_t<UncompressedSize>*    __restrict__  run_data,
	const uint_t<PositionOffsetSize>*  __restrict__  run_start_positions,
	const uint_t<RunIndexSize>*        __restrict__  finer_anchors,
	uint64_t                                         anchoring_period,
	uint64_t                                         position_period,
	size_type_by_index_size<RunIndexSize>            num_anchors,
	size_type_by_index_size<RunIndexSize>            num_element_runs,
	size_type_by_index_size<UncompressedIndexSize>   uncompressed_length)
{
	detail::decompress<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, true, RunIndexSize>(
		decompressed, run_data, run_start_positions, finer_anchors, anchoring_period, position_period, num_anchors, num_element_runs, uncompressed_length);
}

/// LONG segments of sparse runs in ONE launch (round 4; run_length_encoding.hpp, decompress_long_segments): the workgroup that decodes (its share of) a segment first finds
/// the run covering every `fine::sub`-th output of it — a binary search of the segment's run positions per finer anchor, a thread each — keeps those anchors in LDS and decodes
/// the finer segments, a wavefront per finer segment in turn: no launch of the search by itself, no trip of the anchors through memory.
template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned PositionOffsetSize, bool PositionsAreRelative, unsigned RunIndexSize = UncompressedIndexSize>
__global__ void __launch_bounds__(run_expansion::fine::threads) decompress_long_segments(
	uint_t<UncompressedSize>*          __restrict__  decompressed,
	const uint_t<UncompressedSize>*    __restrict__  run_data,
	const uint_t<PositionOffsetSize>*  __restrict__  run_start_positions,
	const uint_t<RunIndexSize>*        __restrict__  position_anchors,
	uint64_t                                         anchoring_period,
	size_type_by_index_size<RunIndexSize>            num_anchors,
	size_type_by_index_size<RunIndexSize>            num_element_runs,
	size_type_by_index_size<UncompressedIndexSize>   uncompressed_length,
	uint32_t                                         segments_per_group)   // >= 1
{
	namespace fine = run_expansion::fine;
	using run_length_encoding::most_fines_per_long_segment;
	using run_length_encoding::finer_segments_per_workgroup;
	__shared__ uint_t<RunIndexSize> finer_anchors[most_fines_per_long_segment + 1];
	const uint64_t n = uncompressed_length, num_runs = num_element_runs;
	const uint64_t group_outputs = anchoring_period * segments_per_group;   // (a group of segments decodes like one segment of as many outputs: run_length_encoding.hpp)
	const uint64_t num_groups = ((uint64_t) num_anchors + segments_per_group - 1) / segments_per_group;
	const uint64_t per_group = (group_outputs + fine::sub - 1) / fine::sub;   // (the wrapper: at most most_fines_per_long_segment)
	const unsigned working_waves = wave64::per_block() < run_expansion::max_waves_per_block ? wave64::per_block() : run_expansion::max_waves_per_block;
	const uint64_t parts = (per_group + finer_segments_per_workgroup - 1) / finer_segments_per_workgroup;
	for (uint64_t work = blockIdx.x; work < num_groups * parts; work += gridDim.x) {
		const uint64_t g = work / parts, part = work - g * parts;
		const uint64_t group_begin = g * group_outputs, group_end = group_begin + group_outputs < n ? group_begin + group_outputs : n;
		const uint64_t fines = (group_end - group_begin + fine::sub - 1) / fine::sub;
		const bool has_next = g + 1 < num_groups;
		// fine::anchors_by_search for this group — a WAVEFRONT per finer anchor, 64 probes a round (fine::last_start_at_most): a thread's binary search over a segment's
		// ~1000 runs was ten dependent round trips at the head of every workgroup's life
		const uint64_t group_anchor = position_anchors[g * segments_per_group];
		const uint64_t next_group_anchor = position_anchors[(has_next ? g + 1 : g) * segments_per_group];
		for (uint64_t j = wave64::index_in_block(); j < fines; j += wave64::per_block()) {
			uint64_t low = group_anchor;
			if (j != 0) {
				// (relative positions restart in every segment of the column: the search is among the runs of the segment the finer anchor's output lies in)
				const uint64_t output = group_begin + j * fine::sub, segment = PositionsAreRelative ? output / anchoring_period : g * segments_per_group;
				uint64_t high = has_next ? next_group_anchor : num_runs - 1;
				if (PositionsAreRelative) {
					const bool a_segment_follows = segment + 1 < (uint64_t) num_anchors;
					const uint64_t this_anchor = position_anchors[segment], following_anchor = position_anchors[a_segment_follows ? segment + 1 : segment];
					low = this_anchor; high = a_segment_follows ? following_anchor : num_runs - 1;
					if (high > low && run_start_positions[high] == 0) { high--; }
				}
				const uint64_t target = PositionsAreRelative ? output - segment * anchoring_period : output;
				low = fine::last_start_at_most(run_start_positions, low, high, target, wave64::lane_id());
			}
			if (wave64::lane_id() == 0) { finer_anchors[j] = (uint_t<RunIndexSize>) low; }
		}
		if (threadIdx.x == 0 && has_next) { finer_anchors[fines] = (uint_t<RunIndexSize>) next_group_anchor; }   // (where the last finer segment's candidates end)
		__syncthreads();
		run_length_encoding::detail::chunks_to_take mine;
		const uint64_t numbering_from = g * per_group;
		const uint64_t part_first = part * finer_segments_per_workgroup;
		const uint64_t part_limit = part_first + finer_segments_per_workgroup < fines ? part_first + finer_segments_per_workgroup : fines;
		mine.given = true; mine.first = numbering_from + part_first + wave64::index_in_block(); mine.stride = working_waves; mine.limit = numbering_from + part_limit;
		mine.anchors_from = numbering_from; mine.outputs_from = group_begin; mine.outputs_until = group_end;
		if (PositionsAreRelative && segments_per_group > 1) { mine.segment_anchors = position_anchors; mine.segment_period = anchoring_period; }
		detail::decompress<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize, false>(
			decompressed, run_data, run_start_positions, finer_anchors, (uint64_t) fine::sub, PositionsAreRelative ? anchoring_period : (uint64_t) fine::sub,
			(size_type_by_index_size<RunIndexSize>) (numbering_from + fines + (has_next ? 1 : 0)), num_element_runs, uncompressed_length, mine);
		__syncthreads();   // (the anchors in LDS are the next group's after this)
	}
}

template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned PositionOffsetSize, bool PositionsAreRelative, unsigned RunIndexSize = UncompressedIndexSize>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;

	launch_config_resolution_params_t(
		device::properties_t            device_properties_,
		device_function::attributes_t   kernel_function_attributes_,
		size_t                          uncompressed_length,
		size_t                          position_anchoring_period,
		optional<shared_memory_size_t>  dynamic_shared_mem_limit = nullopt)
	: parent(device_properties_, kernel_function_attributes_, dynamic_shared_mem_limit)
	{
		if (position_anchoring_period == 0) { throw std::invalid_argument("The anchoring period must be a positive number"); }
		grid_construction_resolution  = warp;                // one wavefront per anchored segment
		length                        = util::div_rounding_up(uncompressed_length, position_anchoring_period);
		length = util::div_rounding_up(length, (size_t) detail::segments_per_wave_chunk(position_anchoring_period, PositionsAreRelative, UncompressedSize));   // batches of short segments
		serialization_option          = auto_maximized;
		quanta.threads_in_block       = warp_size;
		upper_limits.threads_in_block = 256;
		keep_gpu_busy_factor          = 1024;  // in practice: one tile per workgroup — the dispatcher hands tiles out in address order, which HBM likes best
		// 32-bit values, 16-bit positions: 6 resident workgroups per CU are 5-6 points faster than 8 (measured per instantiation, tools/occupancy_sweep.py)
		if (UncompressedSize == 4 && PositionOffsetSize <= 2) { resident_blocks_per_multiprocessor_cap = (unsigned) run_expansion::resident_blocks_cap; }
	}

	launch_config_resolution_params_t(
		device::properties_t device_properties_, size_t uncompressed_length, size_t position_anchoring_period,
		optional<shared_memory_size_t> dynamic_shared_mem_limit = nullopt)
	: launch_config_resolution_params_t(device_properties_,
		device_function_t(decompress<UncompressedIndexSize, UncompressedSize, PositionOffsetSize, PositionsAreRelative, RunIndexSize>).attributes(),
		uncompressed_length, position_anchoring_period, dynamic_shared_mem_limit) { }
};

} // namespace run_position_encoding
} // namespace decompression
} // namespace kernels
} // namespace cuda
