// Run-length decoding for CDNA4:   out[p] = run_data[r(p)],  r(p) = the run covering output p,
// with periodic anchors: segment s (outputs [s*L, (s+1)*L)) begins inside run
// segment_position_anchors[s] at offset segment_position_anchor_intra_run_offsets[s].
//
// Same kernel template name, parameter list and element semantics as the reference's
// src/kernels/decompression/run_length_encoding.cuh:342-374 (per-segment parameters :271-326).
// Execution (see run_expansion.hpp): one WAVEFRONT per anchored segment (or per tile-sized batch of short
// segments), at grid stride;
// the segment is cut into tiles of <= 4096 outputs; per tile the run lengths are read 4 per lane,
// prefix-summed across the wave with DPP (lengths clamped just past the tile length, so 32 bits always
// suffice) to get run START positions, which are marked in an LDS bitmap, ranked, and expanded
// with 16-byte stores.  The run that straddles the end of a tile is carried to the next tile as
// (run index, elements left), so segments of any length work; run lengths must be >= 1, as the
// reference requires.
//
// HBM bytes per output: element + (element + length)/mean_run + anchors/L  (4.19 B for 32-bit values,
// 16-bit lengths, mean run 32, L = 4096 — SURVEY.md §8d, config C4).
#pragma once
#include <hip/hip_runtime.h>

#include "giddy/common.hpp"
#include "giddy/kernels/decompression/run_expansion.hpp"
#include "giddy/kernels/decompression/run_anchors.hpp"
#include "giddy/kernels/launch_config_resolution_params.hpp"
#include "giddy/on_device/batch_pieces.hpp"
#include "giddy/util/math.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace run_length_encoding {

namespace detail {

/// segments a wavefront takes per trip around its outer loop: as many short ones as fill one tile
__host__ __device__ inline uint64_t segments_per_wave_chunk(uint64_t segment_length, unsigned element_size = 4)
{
	// (1-byte columns: a tile is 4 KiB of output, and a wavefront that lives for one of them spends its life on its first loads — several
	// segments in a row, entered once: run_expansion::narrow_chunk_bytes of output per trip)
	if (segment_length >= run_expansion::tile_length) {
		const uint64_t fit = element_size == 1 ? (uint64_t) run_expansion::narrow_chunk_bytes / segment_length : 1;   // (wider columns: one segment — two per trip +0.5 cold, four -10; 2-byte columns with two, round 5: within the noise of +-2.5 — docs/history/r05.md)
		return fit > 1 ? fit : 1;
	}
	// (segments of more than half a tile go two at a time: two tiles with a carry beat one half-empty tile — 52 -> 66 % at 3000)
	uint64_t fit = run_expansion::tile_length / segment_length;
	fit = fit < run_expansion::segments_per_trip_at_least ? run_expansion::segments_per_trip_at_least : fit;
	return run_expansion::whole_64_byte_blocks(fit, segment_length, element_size);
}

} // namespace detail

namespace detail {

/// Which chunks a wavefront takes: those of the whole grid at grid stride (`given` false) — or (decompress_long_segments) `first`, `first + stride`, ... below
/// `limit`, with the anchor arrays beginning at chunk `anchors_from` (they hold one workgroup's segment's finer anchors, in LDS)
/// — and, when the chunks are pieces of a GROUP of the column's segments (any number of outputs, beginning anywhere), chunk `anchors_from` begins at output
/// `outputs_from` and the group ends before `outputs_until`
/// (run positions stored RELATIVE: `segment_anchors` / `segment_period` — the column's own anchors and anchoring period, which its positions count from, when the
/// chunks are pieces of a group of several of its segments)
struct chunks_to_take {
	bool given = false; uint64_t first = 0, stride = 1, limit = 0, anchors_from = 0, outputs_from = 0, outputs_until = ~(uint64_t) 0;
	const void* segment_anchors = nullptr; uint64_t segment_period = 0;
};

/// The kernel's body.  `WithGroups`: with the path for chunks of many runs (groups of runs through LDS: run_expansion.hpp) — without it the kernel
/// is 15 registers lighter and, more to the point, its waits are not the stricter of two paths': with the compressed input NOT in the Infinity
/// Cache it is 10 points faster at mean runs of 32 (the wrapper launches it for columns whose mean run is 16 or more, where no chunk has enough
/// runs for a group anyway).
template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize, bool WithGroups, bool GroupsAlways = false, typename Grid = launch_grid::whole>
__device__ __forceinline__ void decompress(
	uint_t<UncompressedSize>*             __restrict__  decompressed,
	const uint_t<UncompressedSize>*       __restrict__  run_data,
	const uint_t<RunLengthSize>*          __restrict__  run_lengths,
	const uint_t<RunIndexSize>*           __restrict__  segment_position_anchors,
	const uint_t<RunLengthSize>*          __restrict__  segment_position_anchor_intra_run_offsets,
	size_type_by_index_size<UncompressedIndexSize>      segment_length,
	size_type_by_index_size<UncompressedIndexSize>      num_segments,
	size_type_by_index_size<RunIndexSize>               num_element_runs,
	size_type_by_index_size<UncompressedIndexSize>      total_uncompressed_length,
	const chunks_to_take                                mine = chunks_to_take(),
	Grid                                                grid = Grid())   // (launch_grid.hpp: the launch's own, or a batch piece's part of it)
{
	using namespace run_expansion;
	__shared__ tile_entry tables[max_waves_per_block][tile_words];
	using staging = value_staging<UncompressedSize, RunLengthSize>;
	__shared__ alignas(16) uint_t<UncompressedSize> staged_values[max_waves_per_block][staging::allocated];
	uint_t<UncompressedSize>* staged = staged_values[wave64::index_in_block() % max_waves_per_block];   // (idle wavefronts never use it)
	// Chunks of MANY runs (mean runs below ~16): the chunk's run lengths and values come in GROUPS of up to `group_runs` runs — three
	// coalesced 16-byte-per-lane loads, the next group's requested as soon as the current one is in LDS — and the tiles take their
	// descriptors from LDS: no memory round trip between one tile and the next (each tile used to wait for its own batch of descriptors,
	// requested only once the tile before it had been scanned: at mean runs of 2-4 a wavefront spent three quarters of its life waiting).
	using groups = run_groups<UncompressedSize, RunLengthSize, WithGroups>;
	__shared__ alignas(16) uint_t<RunLengthSize> group_lengths_storage[groups::enabled ? max_waves_per_block : 1][groups::enabled ? groups::allocated : 1];
	uint_t<RunLengthSize>* group_lengths = group_lengths_storage[groups::enabled ? wave64::index_in_block() % max_waves_per_block : 0];

	// LDS is provisioned for max_waves_per_block wavefronts (the launch parameters cap the workgroup at 256
	// threads); in a larger workgroup the extra wavefronts sit idle — there is no barrier they could miss
	const unsigned working_waves = wave64::per_block() < max_waves_per_block ? wave64::per_block() : max_waves_per_block;
	if (wave64::index_in_block() >= working_waves) { return; }
	const unsigned lane = wave64::lane_id();
	tile_entry* table = tables[wave64::index_in_block()];
	const uint64_t wave = (uint64_t) grid.block() * working_waves + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) grid.blocks() * working_waves;
	const uint64_t L = segment_length, n = total_uncompressed_length, num_runs = num_element_runs;
	using length_type = typename std::conditional<(RunLengthSize > 4), uint64_t, uint32_t>::type;   // what a run's length is computed with

	// Short segments are taken several at a time: the runs of consecutive segments are consecutive too — anchors are
	// only entry points — so k segments starting at segment s decode exactly like one segment of k * L outputs
	// entered at anchor s.  (A wavefront pays four dependent memory round trips per entry point.)
	const uint64_t segments_per_chunk = detail::segments_per_wave_chunk(L, UncompressedSize);
	const uint64_t num_chunks = (num_segments + segments_per_chunk - 1) / segments_per_chunk;
	// (4- and 8-byte elements only: narrow columns lose 2-4 points to it even with the output on a page boundary — 41-repetition A/B)
	const uint64_t my_first_chunk = UncompressedSize >= 4 ? first_chunk_by_address(decompressed, segments_per_chunk * L * UncompressedSize, working_waves, wave, num_waves) : wave;
	for (uint64_t c = mine.given ? mine.first : my_first_chunk; c < (mine.given ? mine.limit : num_chunks); c += mine.given ? mine.stride : num_waves) {
		const uint64_t s = c * segments_per_chunk;
		const uint64_t segment_start = mine.given ? mine.outputs_from + (s - mine.anchors_from) * L : s * L;
		if (segment_start >= n) { break; }
		const uint64_t chunk_outputs = segments_per_chunk * L;
		const uint64_t column_or_group_end = n < mine.outputs_until ? n : mine.outputs_until;
		const uint64_t segment_end = (segment_start + chunk_outputs < column_or_group_end) ? segment_start + chunk_outputs : column_or_group_end;
		// (both anchors asked for at once — the second at an index that always exists — and ONE wait: a conditional second load is a second round trip, and with the
		// compressed input cold a round trip is 2 microseconds of a wavefront's 17; round 5)
		const bool a_chunk_follows = s + segments_per_chunk < num_segments;
		// (the first run's length arrives with the first batch of lengths: no round trip of its own; how far into it the segment begins: asked for FIRST — with 4- and
		// 8-byte lengths it is a scalar load like the anchors', and behind them it was issued behind their wait)
		length_type skipped_of_first_run = segment_position_anchor_intra_run_offsets[s - mine.anchors_from];
		const uint64_t this_anchor = segment_position_anchors[s - mine.anchors_from], next_anchor = segment_position_anchors[(a_chunk_follows ? s + segments_per_chunk : s) - mine.anchors_from];
		uint64_t first_run = this_anchor;
		length_type left_in_first_run = 0;
		bool entering = true;   // the chunk's first tile: the first run's length is not known yet
		// (a select, not a branch around the use: the compiler sinks a load that only one side of a branch uses back into it — and the second round trip is back)
		const uint64_t through = a_chunk_follows ? next_anchor + 1 : num_runs;
		uint64_t run_limit = through < num_runs ? through : num_runs;   // exclusive

		// groups of runs through LDS (see above) when the chunk has more runs than one batch and the arrays allow 16-byte loads
		// (`GroupsAlways`: the wrapper's launch for dense columns, arrays 16-byte aligned: every chunk by groups — the other path is compiled out)
		const bool use_groups = (GroupsAlways && groups::enabled) || (groups::enabled && run_limit - first_run > runs_per_batch
			&& memory_access::is_aligned(run_lengths, 16) && memory_access::is_aligned(run_data, 16));
		uint64_t group_base = 0, group_end = 0;            // runs [group_base, group_end) are in LDS (group_base a multiple of 8)
		uint64_t next_group_base = ~(uint64_t) 0;          // the group that is on its way (none: ~0)
		typename groups::in_flight coming = { };
		auto request_group = [&](uint64_t base) {          // base: a multiple of 8, below run_limit
			next_group_base = base;
			groups::request(coming, run_lengths, run_data, base, num_runs, lane);
		};
		auto take_group = [&]() {                          // what request_group asked for -> LDS; the following group is requested
			groups::into_lds(coming, group_lengths, staged, lane);
			group_base = next_group_base;
			group_end = group_base + groups::group_runs < run_limit ? group_base + groups::group_runs : run_limit;
			next_group_base = ~(uint64_t) 0;
			if (group_end < run_limit) { request_group((group_end - groups::overlap) / 8 * 8); }
		};
		if (use_groups) { request_group(first_run / 8 * 8); take_group(); }

		// The NEXT tile's first batch of run lengths (and values) is requested before this tile's expansion stores anything: CDNA4
		// retires vector-memory operations in order, so a load issued after a tile's stores is not back before they are
		// acknowledged — a store drain on top of the load's own round trip, per tile (dense tiles, 1 KiB ... 4 KiB of output: the
		// larger part of a wavefront's life).
		uint_t<RunLengthSize> prefetched_lengths[runs_per_lane];
		uint_t<UncompressedSize> prefetched_values[runs_per_lane] = { };   // (read only where values_prefetched says they were loaded; never indeterminate)
		bool prefetched = false, values_prefetched = false;   // (values: always where they travel with the descriptors; else after a dense tile)
		uint32_t tile_len = 0;
		for (uint64_t tile_start = segment_start; tile_start < segment_end && first_run < run_limit; tile_start += tile_len) {
			tile_len = (uint32_t) ((segment_end - tile_start < tile_length) ? segment_end - tile_start : tile_length);
			bool need_carry = tile_start + tile_len < segment_end;
			bool dense = false, values_staged = false;
			if (use_groups && next_group_base <= first_run) { take_group(); }   // (the tile before ended at the old group's end: see the cut below)
			clear(table, lane);

			// Runs are addressed RELATIVE to the tile's first run from here on (32 bits: a tile holds at most tile_length of them), and a
			// batch's loads are unconditional — a lane past the last run re-reads that run and discards it — so that all of them are in
			// flight together: no branch, no wait between one load and the next.
			// (groups: a tile sees the runs that are in LDS — and one batch of them at most while the chunk has runs beyond the group, so that a
			// tile which they do not fill is cut short where dense tiles are: after its first batch, before anything is marked)
			const uint64_t visible_limit = !use_groups ? run_limit : (group_end < run_limit && first_run + runs_per_batch < group_end ? first_run + runs_per_batch : group_end);
			const uint32_t runs_from_here = visible_limit - first_run > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t) (visible_limit - first_run);   // >= 1
			const uint_t<RunLengthSize>* staged_lengths_from_here = group_lengths + (first_run - group_base);
			const uint_t<RunLengthSize>* __restrict__ lengths_from_here = run_lengths + first_run;
			const uint_t<UncompressedSize>* __restrict__ values_from_here = run_data + first_run;
			uint32_t running = 0;      // clamped lengths summed over earlier batches = start of this batch's first run
			uint32_t batch_base = 0;   // index (relative to first_run) of this batch's first run
			uint32_t carried_run = ~0u; length_type carried_left = 0;   // the run that reaches past the tile's end, in the lane that holds it
			do {
				length_type lengths[runs_per_lane]; uint32_t clamped[runs_per_lane]; bool have[runs_per_lane];
				uint_t<UncompressedSize> values[runs_per_lane] = { };
				const uint32_t lane_base = batch_base + lane * runs_per_lane;
				const bool from_prefetch = prefetched && batch_base == 0;   // (wave-uniform)
				const bool stage_prefetched = !staging::with_descriptors && from_prefetch && values_prefetched;   // a dense tile's successor
				if (from_prefetch) {
					#pragma unroll
					for (unsigned j = 0; j < runs_per_lane; j++) { lengths[j] = prefetched_lengths[j]; values[j] = prefetched_values[j]; }
				}
				else if (use_groups) {
					#pragma unroll
					for (unsigned j = 0; j < runs_per_lane; j++) {
						const uint32_t k = lane_base + j < runs_from_here ? lane_base + j : runs_from_here - 1;
						lengths[j] = staged_lengths_from_here[k];   // (the values are in LDS already, at the same offsets)
					}
				}
				else {
					#pragma unroll
					for (unsigned j = 0; j < runs_per_lane; j++) {
						const uint32_t k = lane_base + j < runs_from_here ? lane_base + j : runs_from_here - 1;
						lengths[j] = load_descriptor(lengths_from_here + k);
						if (staging::with_descriptors) { values[j] = values_from_here[k]; }   // the runs' values travel with their lengths
					}
				}
				uint32_t lane_total = 0;
				#pragma unroll
				for (unsigned j = 0; j < runs_per_lane; j++) {
					have[j] = lane_base + j < runs_from_here;
					if (!have[j]) { lengths[j] = 0; }
					if (j == 0 && lane_base == 0) { lengths[0] = entering ? lengths[0] - skipped_of_first_run : left_in_first_run; }
					clamped[j] = lengths[j] > tile_len ? tile_len + 1 : (uint32_t) lengths[j];   // tile_len + 1: "starts past the tile"
					lane_total += clamped[j];
					// (staged for the expansion; runs past the tile's last are harmless)
					if (!use_groups && (staging::with_descriptors || stage_prefetched) && have[j] && lane_base + j < staging::capacity) { staged[lane_base + j] = values[j]; }
				}
				values_staged = values_staged || stage_prefetched;
				uint32_t inclusive = wave64::inclusive_sum(lane_total);
				const uint32_t covered = wave64::read_last_lane(inclusive);
				if (batch_base == 0 && covered < tile_len && covered >= staging::dense_granule && (use_groups ? visible_limit < run_limit : runs_from_here > runs_per_batch)) {   // (lengths of 0 are not allowed: no empty tile)
					// a dense tile (run_expansion.hpp): it ends where this batch's runs end, on a whole 128-byte piece of output
					// (no length was clamped: each is at most `covered`)
					tile_len = covered / staging::dense_granule * staging::dense_granule;
					need_carry = true; dense = true;
				}
				uint32_t start = running + inclusive - lane_total;
				#pragma unroll
				for (unsigned j = 0; j < runs_per_lane; j++) {
					if (have[j] && start - 1u < tile_len - 1u) { mark(table, start); }   // 0 < start < tile_len
					// (a run past the last has length 0 and reaches nowhere)
					const bool reaches_past = need_carry && start <= tile_len && lengths[j] > (length_type) (tile_len - start);
					carried_run = reaches_past ? lane_base + j : carried_run;
					carried_left = reaches_past ? lengths[j] - (length_type) (tile_len - start) : carried_left;
					start += clamped[j];
				}
				running += covered;
				batch_base += runs_per_batch;
			} while ((running < tile_len || (need_carry && running == tile_len)) && batch_base < runs_from_here);

			entering = false;
			// where the next tile enters the runs — and its first batch, on its way before this tile stores
			uint64_t next_first_run = first_run; length_type next_left_in_first_run = left_in_first_run;
			prefetched = false;
			if (need_carry) {
				uint64_t who = wave64::ballot(carried_run != ~0u);
				if (who != 0) {
					unsigned source = (unsigned) __builtin_ctzll(who);
					next_first_run = first_run + wave64::read_lane(carried_run, source);
					next_left_in_first_run = wave64::read_lane(carried_left, source);
				}
				else if (use_groups) {
					// the tile ends exactly where the runs it could see end: the next one begins with the first run it could not see, whole
					// (its length arrives with that tile's first batch, as a chunk's first run's does)
					next_first_run = first_run + runs_from_here;
					entering = true; skipped_of_first_run = 0;
				}
				if (UncompressedSize >= prefetch_from_element_size && !use_groups) {   // (1-byte columns: 16 outputs per lane, the registers are worth more)
					const uint32_t runs_from_there = run_limit - next_first_run > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t) (run_limit - next_first_run);   // >= 1
					uint32_t k[runs_per_lane];
					#pragma unroll
					for (unsigned j = 0; j < runs_per_lane; j++) {
						k[j] = lane * runs_per_lane + j < runs_from_there ? lane * runs_per_lane + j : runs_from_there - 1;
						prefetched_lengths[j] = load_descriptor(run_lengths + next_first_run + k[j]);
					}
					prefetched = true; values_prefetched = staging::with_descriptors || dense;   // (after a dense tile, most likely another)
					if (values_prefetched) {
						#pragma unroll
						for (unsigned j = 0; j < runs_per_lane; j++) { prefetched_values[j] = run_data[next_first_run + k[j]]; }
					}
				}
			}

			const uint32_t marks = rank(table, lane);
			constexpr int staged_stores = staged_store_policy<UncompressedSize, RunLengthSize, false>();
			if (use_groups) {   // the tile's values are in LDS, at the group's offsets
				expand<UncompressedSize, true, true, staged_stores>(decompressed + tile_start, tile_len, table, run_data + first_run, lane, staged + (first_run - group_base), marks + 1);
			}
			else if (staging::dense_only && !dense) {
				expand<UncompressedSize, false>(decompressed + tile_start, tile_len, table, run_data + first_run, lane);
			}
			else if (!staging::with_descriptors && values_staged) {   // a dense tile whose values came with the prefetched lengths
				expand<UncompressedSize, true, true, staged_stores>(decompressed + tile_start, tile_len, table, run_data + first_run, lane, staged, marks + 1);
			}
			else {
				expand<UncompressedSize, true, staging::with_descriptors, staged_stores>(decompressed + tile_start, tile_len, table, run_data + first_run, lane, staged, marks + 1);
			}
			first_run = next_first_run; left_in_first_run = next_left_in_first_run;
		}
	}
}

} // namespace detail

template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize = UncompressedIndexSize>
__global__ void decompress(
	uint_t<UncompressedSize>*             __restrict__  decompressed,
	const uint_t<UncompressedSize>*       __restrict__  run_data,
	const uint_t<RunLengthSize>*          __restrict__  run_lengths,
	const uint_t<RunIndexSize>*           __restrict__  segment_position_anchors,
	const uint_t<RunLengthSize>*          __restrict__  segment_position_anchor_intra_run_offsets,
	size_type_by_index_size<UncompressedIndexSize>      segment_length,
	size_type_by_index_size<UncompressedIndexSize>      num_segments,
	size_type_by_index_size<RunIndexSize>               num_element_runs,
	size_type_by_index_size<UncompressedIndexSize>      total_uncompressed_length)
{
	detail::decompress<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize, true>(decompressed, run_data, run_lengths,
		segment_position_anchors, segment_position_anchor_intra_run_offsets, segment_length, num_segments, num_element_runs, total_uncompressed_length);
}

/// ... with ONLY the groups-of-runs path: the wrapper's launch for columns of mean runs below 16 whose run arrays are 16-byte aligned
template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize = UncompressedIndexSize>
__global__ void decompress_dense_runs(
	uint_t<UncompressedSize>*             __restrict__  decompressed,
	const uint_t<UncompressedSize>*       __restrict__  run_data,
	const uint_t<RunLengthSize>*          __restrict__  run_lengths,
	const uint_t<RunIndexSize>*           __restrict__  segment_position_anchors,
	const uint_t<RunLengthSize>*          __restrict__  segment_position_anchor_intra_run_offsets,
	size_type_by_index_size<UncompressedIndexSize>      segment_length,
	size_type_by_index_size<UncompressedIndexSize>      num_segments,
	size_type_by_index_size<RunIndexSize>               num_element_runs,
	size_type_by_index_size<UncompressedIndexSize>      total_uncompressed_length)
{
	detail::decompress<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize, true, true>(decompressed, run_data, run_lengths,
		segment_position_anchors, segment_position_anchor_intra_run_offsets, segment_length, num_segments, num_element_runs, total_uncompressed_length);
}

/// ... without the groups-of-runs path (detail::decompress): the wrapper's launch for columns of mean runs of 16 or more
template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize = UncompressedIndexSize>
__global__ void decompress_sparse_runs(
	uint_t<UncompressedSize>*             __restrict__  decompressed,
	const uint_t<UncompressedSize>*       __restrict__  run_data,
	const uint_t<RunLengthSize>*          __restrict__  run_lengths,
	const uint_t<RunIndexSize>*           __restrict__  segment_position_anchors,
	const uint_t<RunLengthSize>*          __restrict__  segment_position_anchor_intra_run_offsets,
	size_type_by_index_size<UncompressedIndexSize>      segment_length,
	size_type_by_index_size<UncompressedIndexSize>      num_segments,
	size_type_by_index_size<RunIndexSize>               num_element_runs,
	size_type_by_index_size<UncompressedIndexSize>      total_uncompressed_length)
{
	detail::decompress<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize, false>(decompressed, run_data, run_lengths,
		segment_position_anchors, segment_position_anchor_intra_run_offsets, segment_length, num_segments, num_element_runs, total_uncompressed_length);
}

/// A piece of a batch launch (giddy/batch.hpp; launched as batch::decode_pieces<piece_body<...>>): `decompress`'s parameters out of the piece's record, the body
/// — with or without the groups-of-runs path, as the three kernels above; the wrapper picks by the batch's mean run — on the piece's part of the grid
template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize, bool WithGroups, bool GroupsAlways>
struct piece_body {
	enum : unsigned { max_threads_per_block = 1024 };
	static __device__ __forceinline__ void run(const batch::piece_view piece, const launch_grid::part grid)
	{
		using size_type = size_type_by_index_size<UncompressedIndexSize>;
		detail::decompress<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize, WithGroups, GroupsAlways>(
			piece.get<uint_t<UncompressedSize>*>(0), piece.get<const uint_t<UncompressedSize>*>(1), piece.get<const uint_t<RunLengthSize>*>(2),
			piece.get<const uint_t<RunIndexSize>*>(3), piece.get<const uint_t<RunLengthSize>*>(4),
			piece.get<size_type>(5), piece.get<size_type>(6), piece.get<size_type_by_index_size<RunIndexSize>>(7), piece.get<size_type>(8),
			detail::chunks_to_take(), grid);
	}
};

/// LONG segments of sparse runs in ONE launch (round 4): a workgroup per anchored segment first finds the entry point of every `fine::sub`-th output of
/// ITS segment — run_anchors.hpp, finer_anchors_of_one_segment: the segment's run lengths streamed once, their ends scanned into LDS, a binary search
/// per finer anchor — keeps them in LDS, and goes on to decode the segment, a wavefront per finer segment in turn.  The pre-pass as a launch of its own
/// (fine::anchors_of_segments + decompress_sparse_runs on the finer segments) cost a launch boundary — the decoder's ramp behind the pre-pass's
/// tail: ~20 us of 200 at 2^28 outputs — and a trip of the anchors through memory.  Segments of up to `most_fines` x fine::sub outputs.
enum : unsigned { most_fines_per_long_segment = 64, finer_segments_per_workgroup = 8 };
template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize = UncompressedIndexSize>
__global__ void __launch_bounds__(run_expansion::fine::threads) decompress_long_segments(
	uint_t<UncompressedSize>*             __restrict__  decompressed,
	const uint_t<UncompressedSize>*       __restrict__  run_data,
	const uint_t<RunLengthSize>*          __restrict__  run_lengths,
	const uint_t<RunIndexSize>*           __restrict__  segment_position_anchors,
	const uint_t<RunLengthSize>*          __restrict__  segment_position_anchor_intra_run_offsets,
	size_type_by_index_size<UncompressedIndexSize>      segment_length,
	size_type_by_index_size<UncompressedIndexSize>      num_segments,
	size_type_by_index_size<RunIndexSize>               num_element_runs,
	size_type_by_index_size<UncompressedIndexSize>      total_uncompressed_length,
	uint32_t                                            segments_per_group)   // >= 1: a workgroup's unit is this many consecutive segments, entered at the first one's anchor
{
	namespace fine = run_expansion::fine;
	__shared__ fine::segment_scan_storage<RunLengthSize> storage;
	__shared__ uint_t<RunIndexSize> finer_anchors[most_fines_per_long_segment + 1];
	__shared__ uint_t<RunLengthSize> finer_offsets[most_fines_per_long_segment + 1];
	const uint64_t n = total_uncompressed_length;
	// Anchors are only entry points — the runs of consecutive segments are consecutive — so a GROUP of segments decodes like one segment of as many outputs, of any
	// length and beginning anywhere: its finer anchors are counted from ITS first output (the last finer segment is as long as what is left of the group).
	const uint64_t group_outputs = (uint64_t) segment_length * segments_per_group;
	const uint64_t num_groups = ((uint64_t) num_segments + segments_per_group - 1) / segments_per_group;
	const uint64_t per_group = (group_outputs + fine::sub - 1) / fine::sub;   // (the wrapper: at most most_fines_per_long_segment)
	const unsigned working_waves = wave64::per_block() < run_expansion::max_waves_per_block ? wave64::per_block() : run_expansion::max_waves_per_block;
	// A workgroup decodes at most `finer_segments_per_workgroup` finer segments (two per wavefront: with four — segments of 131 072 — the separate launches were the
	// faster ones): a longer group is shared by several workgroups, each of which scans the group's run lengths for itself (2 bytes per run: kilobytes)
	const uint64_t parts = (per_group + finer_segments_per_workgroup - 1) / finer_segments_per_workgroup;
	for (uint64_t work = blockIdx.x; work < num_groups * parts; work += gridDim.x) {
		const uint64_t g = work / parts, part = work - g * parts;
		const uint64_t group_begin = g * group_outputs, group_end = group_begin + group_outputs < n ? group_begin + group_outputs : n;
		const uint64_t fines = (group_end - group_begin + fine::sub - 1) / fine::sub;
		// (the entry after the group's last: the next group's own anchor — where the last finer segment's runs end; asked for BEFORE the scan of the group's run lengths, not
		// behind its waits)
		const bool has_next = g + 1 < num_groups;
		const uint64_t next_group_at = (has_next ? g + 1 : g) * segments_per_group;
		const uint_t<RunIndexSize> next_group_anchor = segment_position_anchors[next_group_at];
		const uint_t<RunLengthSize> next_group_offset = segment_position_anchor_intra_run_offsets[next_group_at];
		fine::finer_anchors_of_one_segment<RunLengthSize, RunIndexSize>(finer_anchors, finer_offsets, storage,
			run_lengths, segment_position_anchors, segment_position_anchor_intra_run_offsets, g, group_outputs, num_groups, num_element_runs, n, segments_per_group);
		if (threadIdx.x == 0 && has_next) { finer_anchors[fines] = next_group_anchor; finer_offsets[fines] = next_group_offset; }
		__syncthreads();
		detail::chunks_to_take mine;
		const uint64_t numbering_from = g * per_group;   // (the finer segments' numbers only have to be this workgroup's own)
		const uint64_t part_first = part * finer_segments_per_workgroup;
		const uint64_t part_limit = part_first + finer_segments_per_workgroup < fines ? part_first + finer_segments_per_workgroup : fines;
		mine.given = true; mine.first = numbering_from + part_first + wave64::index_in_block(); mine.stride = working_waves; mine.limit = numbering_from + part_limit;
		mine.anchors_from = numbering_from; mine.outputs_from = group_begin; mine.outputs_until = group_end;
		detail::decompress<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize, false>(decompressed, run_data, run_lengths,
			finer_anchors, finer_offsets, (size_type_by_index_size<UncompressedIndexSize>) fine::sub,
			(size_type_by_index_size<UncompressedIndexSize>) (numbering_from + fines + (has_next ? 1 : 0)),   // (as far as the anchors in LDS go)
			num_element_runs, total_uncompressed_length, mine);
		__syncthreads();   // (the anchors in LDS are the next group's after this)
	}
}

template <unsigned UncompressedIndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize = UncompressedIndexSize>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;

	launch_config_resolution_params_t(
		device::properties_t            device_properties_,
		device_function::attributes_t   kernel_function_attributes_,
		size_t                          uncompressed_length,
		size_t                          segment_length,
		optional<shared_memory_size_t>  dynamic_shared_mem_limit = nullopt)
	: parent(device_properties_, kernel_function_attributes_, dynamic_shared_mem_limit)
	{
		if (segment_length == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
		grid_construction_resolution  = warp;                // one wavefront per anchored segment
		length                        = util::div_rounding_up(util::div_rounding_up(uncompressed_length, segment_length),
		                                    (size_t) detail::segments_per_wave_chunk(segment_length, UncompressedSize));
		serialization_option          = auto_maximized;      // (batches of short) segments at grid stride
		quanta.threads_in_block       = warp_size;
		upper_limits.threads_in_block = 256;
		keep_gpu_busy_factor          = 1024;  // in practice: one tile per workgroup — the dispatcher hands tiles out in address order, which HBM likes best
		// 32-bit values, narrow run lengths: 6 resident workgroups per CU are 3-7 points faster than 8 (measured per instantiation, tools/occupancy_sweep.py)
		if (UncompressedSize == 4 && RunLengthSize <= 2) { resident_blocks_per_multiprocessor_cap = (unsigned) run_expansion::resident_blocks_cap; }
	}

	launch_config_resolution_params_t(
		device::properties_t device_properties_, size_t uncompressed_length, size_t segment_length,
		optional<shared_memory_size_t> dynamic_shared_mem_limit = nullopt)
	: launch_config_resolution_params_t(device_properties_,
		device_function_t(decompress<UncompressedIndexSize, UncompressedSize, RunLengthSize, RunIndexSize>).attributes(),
		uncompressed_length, segment_length, dynamic_shared_mem_limit) { }
};

} // namespace run_length_encoding
} // namespace decompression
} // namespace kernels
} // namespace cuda
