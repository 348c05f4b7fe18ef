// Finer anchors for run-encoded columns whose anchored segments are LONG.
//
// The run decoders give a wavefront a whole anchored segment (run_expansion.hpp).  That is the right grain for segments of a few
// thousand outputs; a segment of 65 536 keeps one wavefront writing 256 KiB in a row while its neighbours write theirs a megabyte
// away (65 % of the HBM peak where segments of 4096-8192 reach 84 %), and a column in segments of 2^20 does not even fill the chip
// (26 %).  Anchors are only entry points into the run arrays, so the cure needs no new decoder: BEFORE the decoder runs, find the
// entry point of every `sub` outputs — the run covering output f * sub and, for run-LENGTH encoding, how far into that run it
// lies — and let the same decoder work on segments of `sub`.
//
//   run-position encoding: the runs' starts are stored, so the run covering an output is a binary search over the segment's
//     run_start_positions — one thread per finer anchor (`anchors_by_search`);
//   run-length encoding: a run's start is the sum of the lengths before it.  `sum_run_lengths` adds up every 64 lengths (and every
//     65 536), `scan_tiles` turns the sums into start positions of every 64th run — in a coordinate that only differences within
//     ONE segment are taken of: position(r) - position(anchor run) - intra-run offset is r's start relative to the segment's,
//     whatever the producer did outside that segment (shards sliced out of a column begin mid-run) —, and `anchors_from_sums`
//     searches those, then walks the 64 lengths of the tile it lands in: one thread per finer anchor.
//
// Cost: the lengths are read once more (2 bytes per run) and 8 bytes per 64 runs + one anchor per `sub` outputs are written and
// read back: 3-5 % of the column's traffic at mean runs of 10-30.  Where the reference is: it has no such pass — its kernel is
// what the segment length makes of it (run_length_encoding.cuh:271-326, run_position_encoding.cuh:255-308 take a segment's
// anchor as the only entry point).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>

#include "giddy/common.hpp"
#include "giddy/on_device/wave64.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace run_expansion {
/// The finer anchors' thresholds (measured: profiles/r03_run_long_segments.txt, r04_run_prepass.txt)
enum : unsigned {
	outputs_per_finer_anchor    = 8192,   // segments of 4096-8192 outputs are where the decoders do best
	fused_from_segments         = 512,    // long segments: pre-pass and decoder in one kernel from this many workgroups on (round 5: 2048 left a 2^26-output column in segments of 65 536 — 1024 workgroups — at 50 % where the one kernel does 64)
	fused_up_to_workgroups      = 16384,  // ... and, run lengths, up to this many
	positions_fused_up_to_workgroups = 8192,   // ... run positions: up to this many (16 384 — 2^29 outputs in segments of 32 768, cold: 62.6 % against 70-72 % for the search as a launch of its own; 8192 — 2^28: 73 against 72-75)
	positions_fused_up_to_fines = 4,      // run POSITIONS: the search for finer anchors and the decoder in one kernel for segments of up to this many finer ones
	fused_up_to_fines           = 16,     // run lengths: ... of up to this many (64 — what the kernel's LDS holds the anchors of — measured: eight workgroups each scanning a 524 288-segment's 16 384 run lengths lose 5-7 points to the separate launches)
};
/// Segments at least this long get finer anchors (the wrappers ask); GIDDY_RUN_FINER_ANCHORS_FROM in the environment: another
/// threshold (0: never — measurements).  Default: two finer segments' worth.
/// switches for measurements, read from the environment once per process: GIDDY_RUN_FINE_ONE_LAUNCH=0 — the three-launch pre-pass for
/// run-length columns whatever their shape; GIDDY_RUN_FINE_MIN_MEAN_RUN=m — no finer anchors for run-length columns of shorter mean runs
struct fine_environment {
	bool one_launch = true;
	bool force = false;              // GIDDY_RUN_FINE_FORCE=1: finer anchors whenever the segments allow them
	bool force_one_launch = false;   // GIDDY_RUN_FINE_ONE_LAUNCH=2: ... by the one-launch pre-pass
	uint64_t min_mean_run = 0;
	bool fused = true;               // GIDDY_RUN_FUSED_PREPASS=0: the one-launch pre-pass stays a launch of its own (run lengths)
	uint64_t fused_from = fused_from_segments;   // GIDDY_RUN_FUSED_FROM=n: the one-kernel builds from n workgroups on (tests: 1)
	bool group_all = false;          // GIDDY_RUN_GROUP_ALL=1 (measurements): groups of segments through the long segments' kernel for whole-tile segment lengths too
	static const fine_environment& get()
	{
		static const fine_environment e = [] {
			fine_environment read;
			if (const char* v = std::getenv("GIDDY_RUN_FINE_ONE_LAUNCH")) { read.one_launch = v[0] != '0'; read.force_one_launch = v[0] == '2'; }
			if (const char* v = std::getenv("GIDDY_RUN_FINE_MIN_MEAN_RUN")) { read.min_mean_run = std::strtoull(v, nullptr, 10); }
			if (const char* v = std::getenv("GIDDY_RUN_FINE_FORCE")) { read.force = v[0] == '1'; }
			if (const char* v = std::getenv("GIDDY_RUN_FUSED_PREPASS")) { read.fused = v[0] != '0'; }
			if (const char* v = std::getenv("GIDDY_RUN_GROUP_ALL")) { read.group_all = v[0] == '1'; }
			if (const char* v = std::getenv("GIDDY_RUN_FUSED_FROM")) { read.fused_from = std::strtoull(v, nullptr, 10); }
			return read;
		}();
		return e;
	}
};

inline uint64_t finer_anchors_from()
{
	static const uint64_t from = [] {
		const char* v = std::getenv("GIDDY_RUN_FINER_ANCHORS_FROM");
		const uint64_t given = v ? std::strtoull(v, nullptr, 10) : 2 * (uint64_t) outputs_per_finer_anchor;
		return given == 0 ? ~(uint64_t) 0 : given;
	}();
	return from;
}

/// How many segments of `segment_length` outputs (of `element_size` bytes) make a GROUP for the long segments' one-kernel builds (run_length_encoding.hpp,
/// run_position_encoding.hpp: decompress_long_segments) — about 4 finer segments of `sub` outputs.  Two things decide whether a group size is fast (2^28 outputs,
/// every segment length from 100 to 30 000 tried): a whole number of 64-byte blocks of output (62 % of the HBM peak otherwise, against 72), and finer segments —
/// the last one what is left of the group — that the workgroup's four wavefronts share evenly (33 segments of 1000 — four finer segments and 232 outputs: 59 %).
inline uint64_t segments_per_group_for(uint64_t segment_length, unsigned element_size, uint64_t sub)
{
	const uint64_t L = segment_length;
	uint64_t unit = 64, rest = (L * element_size) % 64;
	while (rest != 0) { const uint64_t t = unit % rest; unit = rest; rest = t; }   // gcd(64, bytes per segment)
	const uint64_t multiple = 64 / unit;   // groups are multiples of this many segments
	if (L * multiple <= sub) { return 4 * sub / (L * multiple) * multiple; }   // (as many as stay within 4 finer segments: the fourth is the one that is short)
	uint64_t per_group = multiple;
	double best = 1e30;
	for (uint64_t k = multiple; k * L <= 8 * sub || k == multiple; k += multiple) {   // (up to 8 finer segments: two turns)
		const uint64_t pieces = (k * L + sub - 1) / sub;
		const double turns = (double) ((pieces + 3) / 4 * 4), work = (double) (k * L) / (double) sub;
		if (turns / work < best - 1e-9) { best = turns / work; per_group = k; }
	}
	return per_group;
}

namespace fine {

enum : unsigned {
	sub = outputs_per_finer_anchor,
	threads = 256,
	runs_per_tile = 64,
	tiles_per_block = 1024,
	runs_per_block = runs_per_tile * tiles_per_block,
	most_runs_per_segment_for_one_launch = 1u << 15,   // (a workgroup takes ~1.5 us per 2048 runs: 16 chunks)
	most_segments_for_one_launch = 1u << 13,
};

/// tile_totals[t] = run_lengths[64 t] + ... + run_lengths[64 t + 63] (runs past the last count 0);
/// block_totals[b] = the sum of tiles [1024 b, 1024 (b + 1)).  `run_lengths` 16-byte aligned.
template <unsigned RunLengthSize>
__global__ void __launch_bounds__(threads) sum_run_lengths(
	uint64_t*                     __restrict__  tile_totals,
	uint64_t*                     __restrict__  block_totals,
	const uint_t<RunLengthSize>*  __restrict__  run_lengths,
	uint64_t                                    num_runs)
{
	constexpr unsigned per_lane = 16 / RunLengthSize;               // lengths in one 16-byte load
	constexpr unsigned lanes_per_tile = runs_per_tile / per_lane;   // 4, 8, 16, 32 lanes add up to a tile
	constexpr unsigned runs_per_load = 64 * per_lane;               // what a wavefront's load instruction covers
	constexpr unsigned loads_per_block = runs_per_block / runs_per_load;
	constexpr unsigned waves = threads / 64;
	using sum_type = typename std::conditional<(RunLengthSize > 2), uint64_t, uint32_t>::type;   // (64 lengths of 16 bits fit 32)
	__shared__ uint64_t wave_totals[waves];
	const unsigned lane = wave64::lane_id(), wave = wave64::index_in_block();
	const uint64_t num_blocks = (num_runs + runs_per_block - 1) / runs_per_block;
	for (uint64_t b = blockIdx.x; b < num_blocks; b += gridDim.x) {
		uint64_t mine = 0;
		for (unsigned i = wave; i < loads_per_block; i += waves) {
			const uint64_t first = b * runs_per_block + (uint64_t) i * runs_per_load + lane * per_lane;
			sum_type sum = 0;
			if (first + per_lane <= num_runs) {
				const auto lengths = memory_access::load_group<uint_t<RunLengthSize>, per_lane>(run_lengths + first);
				#pragma unroll
				for (unsigned j = 0; j < per_lane; j++) { sum += (sum_type) lengths[j]; }
			}
			else {
				for (unsigned j = 0; j < per_lane && first + j < num_runs; j++) { sum += (sum_type) run_lengths[first + j]; }
			}
			mine += sum;
			#pragma unroll
			for (unsigned step = 1; step < lanes_per_tile; step <<= 1) { sum += __shfl_xor(sum, (int) step, 64); }
			if (lane % lanes_per_tile == 0 && first < num_runs) { tile_totals[first / runs_per_tile] = sum; }
		}
		mine = wave64::read_last_lane(wave64::inclusive_sum(mine));
		if (lane == 0) { wave_totals[wave] = mine; }
		__syncthreads();
		if (threadIdx.x == 0) {
			uint64_t total = 0;
			for (unsigned w = 0; w < waves; w++) { total += wave_totals[w]; }
			block_totals[b] = total;
		}
		__syncthreads();
	}
}

/// tile_totals -> in place: the sum of the lengths of all runs before the tile (the tile's first run's start, in the coordinate
/// of the header comment)
template <unsigned TilesPerBlock = tiles_per_block>   // (a template so that every translation unit may hold it)
__global__ void __launch_bounds__(threads) scan_tiles(
	uint64_t*        __restrict__  tile_totals,
	const uint64_t*  __restrict__  block_totals,
	uint64_t                       num_runs)
{
	constexpr unsigned waves = threads / 64, per_thread = TilesPerBlock / threads;
	__shared__ uint64_t table[waves + 1];
	const unsigned lane = wave64::lane_id(), wave = wave64::index_in_block();
	const uint64_t num_tiles = (num_runs + runs_per_tile - 1) / runs_per_tile;
	const uint64_t num_blocks = (num_runs + runs_per_block - 1) / runs_per_block;
	for (uint64_t b = blockIdx.x; b < num_blocks; b += gridDim.x) {
		uint64_t before = 0;
		for (uint64_t k = threadIdx.x; k < b; k += threads) { before += block_totals[k]; }
		before = wave64::read_last_lane(wave64::inclusive_sum(before));
		if (lane == 0) { table[wave] = before; }
		__syncthreads();
		uint64_t base = 0;
		for (unsigned w = 0; w < waves; w++) { base += table[w]; }
		__syncthreads();
		const uint64_t first = b * TilesPerBlock + (uint64_t) threadIdx.x * per_thread;
		uint64_t totals[per_thread], sum = 0;
		#pragma unroll
		for (unsigned j = 0; j < per_thread; j++) { totals[j] = first + j < num_tiles ? tile_totals[first + j] : 0; sum += totals[j]; }
		const uint64_t inclusive = wave64::inclusive_sum(sum);
		if (lane == 63) { table[wave] = inclusive; }
		__syncthreads();
		uint64_t start = base + inclusive - sum;
		for (unsigned w = 0; w < wave; w++) { start += table[w]; }
		#pragma unroll
		for (unsigned j = 0; j < per_thread; j++) {
			if (first + j < num_tiles) { tile_totals[first + j] = start; }
			start += totals[j];
		}
		__syncthreads();
	}
}

namespace detail {

/// Walks the lengths of runs [from, ...) while their starts stay at or before `target`: in: `start` = the start of run `from`;
/// out: the last run among [from, limit) that starts at or before `target`, and its start.  (`from` a multiple of a 16-byte
/// group when `run_lengths` is 16-byte aligned.)
template <unsigned RunLengthSize>
__device__ __forceinline__ void walk(const uint_t<RunLengthSize>* __restrict__ run_lengths, uint64_t& run, uint64_t& start, uint64_t limit, uint64_t target, uint64_t num_runs)
{
	constexpr unsigned per_group = 16 / RunLengthSize;
	bool done = false;
	while (!done && run < limit) {
		if (run % per_group == 0 && run + per_group <= num_runs) {
			const auto lengths = memory_access::load_group<uint_t<RunLengthSize>, per_group>(run_lengths + run);
			#pragma unroll
			for (unsigned j = 0; j < per_group; j++) {
				const uint64_t next = start + (uint64_t) lengths[j];
				if (!done && run + 1 < limit && next <= target) { start = next; run++; } else { done = true; }
			}
		}
		else {
			const uint64_t next = start + (uint64_t) run_lengths[run];
			if (run + 1 < limit && next <= target) { start = next; run++; } else { done = true; }
		}
	}
}

} // namespace detail

/// Run-length encoding: one thread per finer segment f (outputs [f * sub, ...)): fine_anchors[f] = the run covering its first
/// output, fine_offsets[f] = how far into that run.  `tile_starts`: what scan_tiles left.
template <unsigned RunLengthSize, unsigned RunIndexSize>
__global__ void __launch_bounds__(threads) anchors_from_sums(
	uint_t<RunIndexSize>*         __restrict__  fine_anchors,
	uint_t<RunLengthSize>*        __restrict__  fine_offsets,
	const uint64_t*               __restrict__  tile_starts,
	const uint_t<RunLengthSize>*  __restrict__  run_lengths,
	const uint_t<RunIndexSize>*   __restrict__  position_anchors,
	const uint_t<RunLengthSize>*  __restrict__  intra_run_anchor_offsets,
	uint64_t                                    segment_length,
	uint64_t                                    num_segments,
	uint64_t                                    num_runs,
	uint64_t                                    length)
{
	const uint64_t per_segment = segment_length / sub, num_fine = (length + sub - 1) / sub;
	for (uint64_t f = (uint64_t) blockIdx.x * threads + threadIdx.x; f < num_fine; f += (uint64_t) gridDim.x * threads) {
		const uint64_t s = f / per_segment, j = f % per_segment;
		const uint64_t anchor = position_anchors[s];
		const uint64_t offset = intra_run_anchor_offsets[s];
		if (j == 0) { fine_anchors[f] = (uint_t<RunIndexSize>) anchor; fine_offsets[f] = (uint_t<RunLengthSize>) offset; continue; }
		const uint64_t last_run = s + 1 < num_segments ? (uint64_t) position_anchors[s + 1] : num_runs - 1;   // the last run the segment can touch
		// the anchor run's start: its tile's, plus the lengths between
		uint64_t run = anchor / runs_per_tile * runs_per_tile, start = tile_starts[anchor / runs_per_tile];
		detail::walk<RunLengthSize>(run_lengths, run, start, anchor + 1, ~(uint64_t) 0, num_runs);   // (walks to `anchor` itself)
		const uint64_t target = start + offset + j * sub;   // where output f * sub lies, in the coordinate of tile_starts
		// the last tile among the segment's that starts at or before the target (the anchor's does)
		uint64_t low = anchor / runs_per_tile, high = last_run / runs_per_tile;
		while (low < high) {
			const uint64_t middle = low + (high - low + 1) / 2;
			if (tile_starts[middle] <= target) { low = middle; } else { high = middle - 1; }
		}
		run = low * runs_per_tile; start = tile_starts[low];
		const uint64_t limit = (low + 1) * runs_per_tile < last_run + 1 ? (low + 1) * runs_per_tile : last_run + 1;
		detail::walk<RunLengthSize>(run_lengths, run, start, limit, target, num_runs);
		fine_anchors[f] = (uint_t<RunIndexSize>) run;
		fine_offsets[f] = (uint_t<RunLengthSize>) (target - start);
	}
}


/// The last run of [low, high] that starts at or before `target` (run `low` does) — by a WAVEFRONT: all 64 lanes call it with the same arguments and probe 64 places a
/// round, so 4096 candidates are two dependent round trips where a thread's binary search is twelve.
template <typename Position>
__device__ __forceinline__ uint64_t last_start_at_most(const Position* __restrict__ starts, uint64_t low, uint64_t high, uint64_t target, unsigned lane)
{
	while (high > low) {
		const uint64_t span = high - low, step = (span + 63) / 64;   // probe k looks at run low + min(span, (k + 1) * step): the 64th at `high`
		auto probe = [&](uint64_t k) { return low + (k * step < span ? k * step : span); };
		const unsigned at_most = (unsigned) __builtin_popcountll(wave64::ballot((uint64_t) starts[probe(lane + 1)] <= target));   // (ones, then zeros: the starts ascend)
		// (probe(at_most + 1) > probe(at_most) whenever at_most < 64: were both clamped to `high`, run `high` would start at or before the target and every probe with it.
		// No `max(new_high, new_low)` here: hipcc 7.2 compiles that uniform 64-bit select to an s_cselect on a stale SCC — tools/experiments/search_probe.hip caught it)
		const uint64_t new_low = probe(at_most), new_high = at_most == 64 ? high : probe(at_most + 1) - 1;
		low = new_low; high = new_high;
	}
	return low;
}

/// Run-length encoding, ONE launch (round 4): a workgroup per anchored segment streams the segment's run lengths ONCE — chunks of
/// threads x (16 / RunLengthSize) runs as 16-byte loads, the next chunk requested before this one is worked on —, scans them (in-lane
/// sums, a 64-lane DPP scan, the wavefronts' totals through LDS), leaves the runs' END positions relative to the segment's start in
/// LDS, and the finer anchors whose outputs (multiples of `sub`) fall among the chunk's runs are found there by binary search, a thread
/// per anchor — however long a run is, a finer anchor costs the same.  Positions are differences within the segment (its anchor run
/// begins `intra_run_anchor_offsets[s]` before it), as in anchors_from_sums; nothing is read of other segments' runs but the next
/// segment's anchor, and no scratch beyond the anchors themselves is written.  The three-launch pass above remains for columns
/// whose segments hold so many runs each (tens of thousands) that a workgroup per segment would leave the chip idle.
/// (the work for ONE segment, by the whole workgroup; `anchors_out` / `offsets_out`: the segment's finer anchors — in global memory, or in LDS
/// when the same workgroup goes on to decode the segment: run_length_encoding.hpp, decompress_long_segments)
template <unsigned RunLengthSize> struct segment_scan_storage {
	enum : unsigned { per_lane = 16 / RunLengthSize, runs_per_chunk = threads * per_lane, waves = threads / 64 };
	// end of each of the chunk's runs, counted from where the CHUNK's first run begins (1-2-byte lengths: a chunk's total fits 32 bits — half the LDS, twice the workgroups per CU)
	using end_type = typename std::conditional<(RunLengthSize <= 2), uint32_t, uint64_t>::type;
	end_type ends[runs_per_chunk];
	uint64_t wave_totals[waves];
};

template <unsigned RunLengthSize, unsigned RunIndexSize>
__device__ __forceinline__ void finer_anchors_of_one_segment(
	uint_t<RunIndexSize>* anchors_out, uint_t<RunLengthSize>* offsets_out, segment_scan_storage<RunLengthSize>& storage,
	const uint_t<RunLengthSize>*  __restrict__  run_lengths,
	const uint_t<RunIndexSize>*   __restrict__  position_anchors,
	const uint_t<RunLengthSize>*  __restrict__  intra_run_anchor_offsets,
	uint64_t s, uint64_t segment_length, uint64_t num_segments, uint64_t num_runs, uint64_t length,
	uint64_t anchor_stride = 1)   // (decompress_long_segments on GROUPS of segments: "segment" s is the caller's segments s * stride ..., its anchor their first's)
{
	using storage_type = segment_scan_storage<RunLengthSize>;
	constexpr unsigned per_lane = storage_type::per_lane, runs_per_chunk = storage_type::runs_per_chunk, waves = storage_type::waves;
	using length_type = uint_t<RunLengthSize>;
	using end_type = typename storage_type::end_type;
	end_type* const ends = storage.ends;
	uint64_t* const wave_totals = storage.wave_totals;
	const unsigned lane = wave64::lane_id(), wave = wave64::index_in_block();

	auto load = [&](uint64_t first, uint64_t from, uint64_t last) {   // this thread's group of the chunk: lengths of runs outside [from, last] count 0
		memory_access::group<length_type, per_lane> lengths;
		if (first >= from && first + per_lane <= last + 1) { lengths = memory_access::load_group<length_type, per_lane>(run_lengths + first); }   // (last < num_runs)
		else {
			#pragma unroll
			for (unsigned j = 0; j < per_lane; j++) { lengths[j] = (first + j < num_runs && first + j >= from && first + j <= last) ? run_lengths[first + j] : (length_type) 0; }
		}
		return lengths;
	};

	// (the next segment's anchor asked for with this one's — at an index that always exists — and selected: loaded under the condition it is a round trip of its own)
	const bool a_segment_follows = s + 1 < num_segments;
	const uint64_t anchor = position_anchors[s * anchor_stride], offset = intra_run_anchor_offsets[s * anchor_stride];
	const uint64_t next_anchor = position_anchors[(a_segment_follows ? s + 1 : s) * anchor_stride];
	const uint64_t last_run = a_segment_follows ? next_anchor : num_runs - 1;   // the last run the segment can touch
	const uint64_t segment_outputs = (s + 1) * segment_length <= length ? segment_length : length - s * segment_length;
	const uint64_t fines = (segment_outputs + sub - 1) / sub;   // finer anchors of this segment; number 0 is the segment's own
	if (threadIdx.x == 0) { anchors_out[0] = (uint_t<RunIndexSize>) anchor; offsets_out[0] = (uint_t<RunLengthSize>) offset; }
	if (fines <= 1) { return; }

	const uint64_t chunk0 = anchor / per_lane * per_lane;   // 16-byte groups: the first chunk begins at the group holding the anchor run
	uint64_t base = 0;                // where the chunk's first run begins, counted from where the anchor run begins
	uint64_t last_start = 0;          // ... where the last run seen begins
	uint64_t next_fine = 1;           // the first finer anchor not yet written
	auto ahead = load(chunk0 + (uint64_t) threadIdx.x * per_lane, anchor, last_run);
	for (uint64_t chunk = chunk0; chunk <= last_run && next_fine < fines; chunk += runs_per_chunk) {
		const uint64_t first = chunk + (uint64_t) threadIdx.x * per_lane;
		const auto lengths = ahead;
		if (chunk + runs_per_chunk <= last_run) { ahead = load(first + runs_per_chunk, anchor, last_run); }
		uint64_t inclusive[per_lane], sum = 0;
		#pragma unroll
		for (unsigned j = 0; j < per_lane; j++) { sum += (uint64_t) lengths[j]; inclusive[j] = sum; }
		const uint64_t scanned = wave64::inclusive_sum(sum);
		if (lane == 63) { wave_totals[wave] = scanned; }
		__syncthreads();
		uint64_t before = scanned - sum, chunk_total = 0;
		#pragma unroll
		for (unsigned w = 0; w < waves; w++) { before += w < wave ? wave_totals[w] : 0; chunk_total += wave_totals[w]; }
		#pragma unroll
		for (unsigned j = 0; j < per_lane; j++) { ends[threadIdx.x * per_lane + j] = (end_type) (before + inclusive[j]); }
		__syncthreads();
		// the finer anchors whose first output lies among this chunk's runs: output j * sub of the segment is at offset + j * sub from where the anchor run begins
		const uint64_t chunk_end = base + chunk_total;
		const uint64_t covered = chunk_end > offset ? (chunk_end - offset + sub - 1) / sub : 0;   // finer anchors j < covered begin before chunk_end
		const uint64_t until = covered < fines ? covered : fines;
		for (uint64_t j = next_fine + threadIdx.x; j < until; j += threads) {
			const uint64_t target = offset + j * sub - base;   // (counted from the chunk's first run, as `ends` are)
			unsigned low = 0, high = runs_per_chunk - 1;   // the first run of the chunk that ends beyond the target
			while (low < high) {
				const unsigned middle = (low + high) / 2;
				if ((uint64_t) ends[middle] > target) { high = middle; } else { low = middle + 1; }
			}
			const uint64_t start = low == 0 ? 0 : (uint64_t) ends[low - 1];
			anchors_out[j] = (uint_t<RunIndexSize>) (chunk + low);
			offsets_out[j] = (uint_t<RunLengthSize>) (target - start);
		}
		if (until > next_fine) { next_fine = until; }
		// (where the segment's last run begins: what is left of the segment is pinned there if the lengths fall short of the segment)
		if (chunk + runs_per_chunk > last_run) { const unsigned at = (unsigned) (last_run - chunk); last_start = base + (at == 0 ? 0 : (uint64_t) ends[at - 1]); }
		base = chunk_end;
		__syncthreads();   // (`ends` and `wave_totals` are rewritten by the next chunk)
	}
	// lengths that do not reach the segment's end (a malformed column): the rest enters at the last run, as the three-launch pass's walk would
	for (uint64_t j = next_fine + threadIdx.x; j < fines; j += threads) {
		anchors_out[j] = (uint_t<RunIndexSize>) last_run;
		offsets_out[j] = (uint_t<RunLengthSize>) (offset + j * sub - last_start);
	}
}

template <unsigned RunLengthSize, unsigned RunIndexSize>
__global__ void __launch_bounds__(threads) anchors_of_segments(
	uint_t<RunIndexSize>*         __restrict__  fine_anchors,
	uint_t<RunLengthSize>*        __restrict__  fine_offsets,
	const uint_t<RunLengthSize>*  __restrict__  run_lengths,
	const uint_t<RunIndexSize>*   __restrict__  position_anchors,
	const uint_t<RunLengthSize>*  __restrict__  intra_run_anchor_offsets,
	uint64_t                                    segment_length,
	uint64_t                                    num_segments,
	uint64_t                                    num_runs,
	uint64_t                                    length)
{
	__shared__ segment_scan_storage<RunLengthSize> storage;
	const uint64_t per_segment = segment_length / sub;
	for (uint64_t s = blockIdx.x; s < num_segments; s += gridDim.x) {
		finer_anchors_of_one_segment<RunLengthSize, RunIndexSize>(fine_anchors + s * per_segment, fine_offsets + s * per_segment, storage,
			run_lengths, position_anchors, intra_run_anchor_offsets, s, segment_length, num_segments, num_runs, length);
	}
}

/// Run-position encoding: one thread per finer segment f: fine_anchors[f] = the last run of its segment that starts at or before
/// output f * sub (the segment's anchor run is taken to start at the segment's start, as the decoder takes it).
template <unsigned PositionOffsetSize, bool PositionsAreRelative, unsigned RunIndexSize>
__global__ void __launch_bounds__(threads) anchors_by_search(
	uint_t<RunIndexSize>*              __restrict__  fine_anchors,
	const uint_t<PositionOffsetSize>*  __restrict__  run_start_positions,
	const uint_t<RunIndexSize>*        __restrict__  position_anchors,
	uint64_t                                         segment_length,
	uint64_t                                         num_segments,
	uint64_t                                         num_runs,
	uint64_t                                         length)
{
	const uint64_t per_segment = segment_length / sub, num_fine = (length + sub - 1) / sub;
	for (uint64_t f = (uint64_t) blockIdx.x * threads + threadIdx.x; f < num_fine; f += (uint64_t) gridDim.x * threads) {
		const uint64_t s = f / per_segment, j = f % per_segment;
		uint64_t low = position_anchors[s];
		if (j != 0) {
			uint64_t high = s + 1 < num_segments ? (uint64_t) position_anchors[s + 1] : num_runs - 1;
			// (relative positions: the next segment's anchor run, if it starts exactly there, says 0 — it is none of this segment's)
			if (PositionsAreRelative && high > low && run_start_positions[high] == 0) { high--; }
			const uint64_t target = (PositionsAreRelative ? 0 : s * segment_length) + j * sub;
			while (low < high) {
				const uint64_t middle = low + (high - low + 1) / 2;
				if ((uint64_t) run_start_positions[middle] <= target) { low = middle; } else { high = middle - 1; }
			}
		}
		fine_anchors[f] = (uint_t<RunIndexSize>) low;
	}
}

} // namespace fine
} // namespace run_expansion
} // namespace decompression
} // namespace kernels
} // namespace cuda
