// Device-side compressors for the two run encodings (formats: libgiddy_amd/formats.py::run_length_encoding,
// ::run_position_encoding; reference: run_length_encoding.cuh:271-326, run_position_encoding.cuh:255-308).
//
// A run starts wherever an element differs from its predecessor.  Element level (three launches, compression/common.hpp):
// run starts per tile -> offsets -> write.  Run-position encoding is finished by that: run k's value, its start position
// (absolute, or relative to its segment) and, for every segment, the index of the run covering its first element.
// Run-LENGTH encoding goes on at the RUN level, because a length field holds at most `max_run_length`: a longer run is cut into
// pieces of that length (formats._runs_of), so the final run index of a run depends on the runs before it — pieces per run ->
// offsets -> write values, lengths and the anchors (run index + offset into the run) of the segments that begin inside each run.
#pragma once
#include "giddy/kernels/compression/common.hpp"

namespace cuda {
namespace kernels {
namespace compression {
namespace runs {

/// tile_totals[t] = number of runs that start in tile t
template <unsigned IndexSize, unsigned UncompressedSize>
__global__ void __launch_bounds__(threads_per_block) count_run_starts(
	uint64_t*                        __restrict__  tile_totals,
	const uint_t<UncompressedSize>*  __restrict__  input,
	size_type_by_index_size<IndexSize>             length)
{
	__shared__ uint64_t table[waves_per_block + 1];
	const uint64_t n = length, num_tiles = (n + tile_elements - 1) / tile_elements;
	for (uint64_t t = blockIdx.x; t < num_tiles; t += gridDim.x) {
		const uint64_t first = t * tile_elements + (uint64_t) threadIdx.x * elements_per_thread;
		uint64_t mine = 0;
		if (first < n) {
			uint_t<UncompressedSize> previous = first == 0 ? (uint_t<UncompressedSize>) ~input[0] : input[first - 1];
			uint_t<UncompressedSize> values[elements_per_thread];
			load_own_elements(values, input, first, n, previous);
			#pragma unroll
			for (unsigned j = 0; j < elements_per_thread; j++) {
				mine += (first + j < n) && values[j] != previous;
				previous = values[j];
			}
		}
		const uint64_t total = block_sum(mine, table);
		if (threadIdx.x == 0) { tile_totals[t] = total; }
	}
}

/// Run k (in order of position): run_starts[k] = its first position; optionally (not null) run_data[k] = its value,
/// run_start_positions[k] = its start (relative: modulo the segment length), position_anchors[s] = the run covering s * L.
template <unsigned IndexSize, unsigned UncompressedSize, unsigned PositionSize, bool PositionsAreRelative, unsigned RunIndexSize>
__global__ void __launch_bounds__(threads_per_block) write_run_starts(
	uint_t<IndexSize>*               __restrict__  run_starts,
	uint_t<UncompressedSize>*        __restrict__  run_data,
	uint_t<PositionSize>*            __restrict__  run_start_positions,
	uint_t<RunIndexSize>*            __restrict__  position_anchors,
	const uint_t<UncompressedSize>*  __restrict__  input,
	const uint64_t*                  __restrict__  tile_offsets,
	size_type_by_index_size<IndexSize>             segment_length,
	size_type_by_index_size<IndexSize>             length,
	uint64_t                                       capacity)   // runs the per-run arrays have room for: runs beyond it are counted, not stored
{
	__shared__ uint64_t table[waves_per_block + 1];
	const uint64_t n = length, L = segment_length, num_tiles = (n + tile_elements - 1) / tile_elements;
	for (uint64_t t = blockIdx.x; t < num_tiles; t += gridDim.x) {
		const uint64_t first = t * tile_elements + (uint64_t) threadIdx.x * elements_per_thread;
		uint_t<UncompressedSize> values[elements_per_thread];
		uint_t<UncompressedSize> previous = 0;
		uint64_t mine = 0;
		if (first < n) {
			previous = first == 0 ? (uint_t<UncompressedSize>) ~input[0] : input[first - 1];
			uint_t<UncompressedSize> last = previous;
			load_own_elements(values, input, first, n, previous);
			#pragma unroll
			for (unsigned j = 0; j < elements_per_thread; j++) {
				mine += (first + j < n) && values[j] != last;
				last = values[j];
			}
		}
		uint64_t tile_total;
		uint64_t k = tile_offsets[t] + block_exclusive_sum(mine, table, tile_total);   // runs that start before this thread's elements
		if (first < n) {
			uint64_t until_anchor = L == 0 ? ~(uint64_t) 0 : (L - first % L) % L;
			#pragma unroll
			for (unsigned j = 0; j < elements_per_thread; j++) {
				const uint64_t i = first + j;
				if (i < n) {
					if (values[j] != previous) {
						if (k < capacity) {
							if (run_starts != nullptr) { run_starts[k] = (uint_t<IndexSize>) i; }
							if (run_data != nullptr) { run_data[k] = values[j]; }
							if (run_start_positions != nullptr) { run_start_positions[k] = (uint_t<PositionSize>) (PositionsAreRelative ? i % L : i); }
						}
						k++;
					}
					if (until_anchor == 0) {
						if (position_anchors != nullptr) { position_anchors[i / L] = (uint_t<RunIndexSize>) (k - 1); }
						until_anchor = L;
					}
					until_anchor--;
					previous = values[j];
				}
			}
		}
	}
}

/// run_tile_totals[t] = pieces of the natural runs [t * tile_elements, (t + 1) * tile_elements): ceil(length / max_run_length) each.
/// `num_natural_runs` is read from the device (what the element-level offsets pass left); tiles beyond it count 0.
template <unsigned IndexSize>
__global__ void __launch_bounds__(threads_per_block) count_pieces(
	uint64_t*                 __restrict__  run_tile_totals,
	const uint_t<IndexSize>*  __restrict__  run_starts,
	const uint64_t*           __restrict__  num_natural_runs,
	uint64_t                                num_run_tiles,
	size_type_by_index_size<IndexSize>      length,
	uint64_t                                max_run_length,
	uint64_t                                starts_capacity)   // entries of run_starts that were stored (write_run_starts' capacity)
{
	__shared__ uint64_t table[waves_per_block + 1];
	const uint64_t runs = *num_natural_runs < starts_capacity ? *num_natural_runs : starts_capacity, n = length;
	for (uint64_t t = blockIdx.x; t < num_run_tiles; t += gridDim.x) {
		const uint64_t first = t * tile_elements + (uint64_t) threadIdx.x * elements_per_thread;
		uint64_t mine = 0;
		for (unsigned j = 0; j < elements_per_thread && first + j < runs; j++) {
			const uint64_t start = run_starts[first + j], end = first + j + 1 < runs ? (uint64_t) run_starts[first + j + 1] : n;
			mine += (end - start - 1) / max_run_length + 1;   // (runs are not empty; no overflow for 64-bit length fields)
		}
		const uint64_t total = block_sum(mine, table);
		if (threadIdx.x == 0) { run_tile_totals[t] = total; }
	}
}

/// The run-length format from the natural runs: values and lengths of the pieces, and for every segment that begins inside a
/// natural run the piece covering its first element and the offset into that piece.
template <unsigned IndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize>
__global__ void __launch_bounds__(threads_per_block) write_pieces(
	uint_t<UncompressedSize>*        __restrict__  run_data,
	uint_t<RunLengthSize>*           __restrict__  run_lengths,
	uint_t<RunIndexSize>*            __restrict__  position_anchors,
	uint_t<RunLengthSize>*           __restrict__  intra_run_anchor_offsets,
	const uint_t<UncompressedSize>*  __restrict__  input,
	const uint_t<IndexSize>*         __restrict__  run_starts,
	const uint64_t*                  __restrict__  num_natural_runs,
	const uint64_t*                  __restrict__  run_tile_offsets,
	uint64_t                                       num_run_tiles,
	size_type_by_index_size<IndexSize>             segment_length,
	size_type_by_index_size<IndexSize>             length,
	uint64_t                                       max_run_length,
	uint64_t                                       starts_capacity,
	uint64_t                                       capacity)   // pieces run_data / run_lengths have room for
{
	__shared__ uint64_t table[waves_per_block + 1];
	const uint64_t runs = *num_natural_runs < starts_capacity ? *num_natural_runs : starts_capacity, n = length, L = segment_length, M = max_run_length;
	for (uint64_t t = blockIdx.x; t < num_run_tiles; t += gridDim.x) {
		const uint64_t first = t * tile_elements + (uint64_t) threadIdx.x * elements_per_thread;
		uint64_t mine = 0;
		for (unsigned j = 0; j < elements_per_thread && first + j < runs; j++) {
			const uint64_t start = run_starts[first + j], end = first + j + 1 < runs ? (uint64_t) run_starts[first + j + 1] : n;
			mine += (end - start - 1) / M + 1;
		}
		uint64_t tile_total;
		uint64_t piece = run_tile_offsets[t] + block_exclusive_sum(mine, table, tile_total);
		for (unsigned j = 0; j < elements_per_thread && first + j < runs; j++) {
			const uint64_t start = run_starts[first + j], end = first + j + 1 < runs ? (uint64_t) run_starts[first + j + 1] : n;
			const uint_t<UncompressedSize> value = input[start];
			const uint64_t pieces = (end - start - 1) / M + 1;
			for (uint64_t p = 0; p < pieces && piece + p < capacity; p++) {
				run_data[piece + p] = value;
				const uint64_t left = end - start - p * M;
				run_lengths[piece + p] = (uint_t<RunLengthSize>) (left < M ? left : M);
			}
			if (L != 0) {
				for (uint64_t s = (start + L - 1) / L; s * L < end; s++) {   // the segments whose first element lies in this run
					const uint64_t into_run = s * L - start;
					position_anchors[s] = (uint_t<RunIndexSize>) (piece + into_run / M);
					intra_run_anchor_offsets[s] = (uint_t<RunLengthSize>) (into_run % M);
				}
			}
			piece += pieces;
		}
	}
}

} // namespace runs
} // namespace compression
} // namespace kernels
} // namespace cuda
