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

#ifndef GIDDY_RUN_FINE_SUB
#define GIDDY_RUN_FINE_SUB 8192   // outputs per finer anchor: segments of 4096-8192 are where the decoders do best (profiles/r03_run_long_segments.txt)
#endif

namespace cuda {
namespace kernels {
namespace decompression {
namespace run_expansion {
/// Segments at least this long get finer anchors (the wrappers ask); GIDDY_RUN_FINER_ANCHORS_FROM in the environment: another
/// threshold (0: never — measurements).  Default: two finer segments' worth.
inline uint64_t finer_anchors_from()
{
	static const uint64_t from = [] {
		const char* v = std::getenv("GIDDY_RUN_FINER_ANCHORS_FROM");
		const uint64_t given = v ? std::strtoull(v, nullptr, 10) : 2 * (uint64_t) GIDDY_RUN_FINE_SUB;
		return given == 0 ? ~(uint64_t) 0 : given;
	}();
	return from;
}

namespace fine {

enum : unsigned {
	sub = GIDDY_RUN_FINE_SUB,
	threads = 256,
	runs_per_tile = 64,
	tiles_per_block = 1024,
	runs_per_block = runs_per_tile * tiles_per_block,
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
