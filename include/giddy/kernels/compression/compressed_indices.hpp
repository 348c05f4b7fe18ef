// Device-side writer of COMPRESSED patch positions: ascending, distinct positions -> runs of (baseline, offset width, length, start in
// a byte pool), a new run at every multiple of `anchoring_period` patches — the arrays the compressed-indices scatter and the fused
// decoders read (format: libgiddy_amd/formats.py::compressed_indices; reference: compressed_indices_scatter.cuh:131-217).
// The writer's run formation is greedy and sequential INSIDE an anchored block — a run's offset width is the width class (8 / 16 / 32 /
// 64 bits) of its second position's offset from its first, and it grows while the offsets stay within that class and the length fits
// the run-length type — and blocks are independent: a THREAD walks a block (patches are a thousandth of a column; this is not a
// bandwidth problem), twice: once to count runs and pool bytes (-> offsets across blocks, compression/common.hpp), once to write.
// Every run's offsets start on an OutputIndexSize-aligned byte of the pool (:198-201); the pool ends on a multiple of 8.
#pragma once
#include "giddy/kernels/compression/common.hpp"

namespace cuda {
namespace kernels {
namespace compression {
namespace compressed_indices {

__device__ __forceinline__ unsigned width_class(uint64_t offset)
{
	return offset < (1ull << 8) ? 8 : offset < (1ull << 16) ? 16 : offset < (1ull << 32) ? 32 : 64;
}

/// the run that starts at patch `i` of a block ending (exclusive) at `end`: its length and offset width
template <unsigned OutputIndexSize>
__device__ __forceinline__ void next_run(const uint_t<OutputIndexSize>* __restrict__ positions, uint64_t i, uint64_t end, uint64_t longest,
	uint64_t& length, unsigned& width)
{
	const uint64_t base = positions[i];
	const uint64_t second = i + 1 < end ? i + 1 : end - 1;
	width = width_class((uint64_t) positions[second] - base);
	uint64_t j = i + 1;
	while (j < end && j - i < longest && width_class((uint64_t) positions[j] - base) <= width) { j++; }
	length = j - i;
}

/// block_runs[a], block_bytes[a] = runs and (aligned) pool bytes of anchored block a
template <unsigned OutputIndexSize, unsigned InputIndexSize, unsigned RunLengthSize>
__global__ void __launch_bounds__(threads_per_block) count_runs(
	uint64_t*                        __restrict__  block_runs,
	uint64_t*                        __restrict__  block_bytes,
	const uint_t<OutputIndexSize>*   __restrict__  positions,
	size_type_by_index_size<InputIndexSize>        num_positions,
	uint_t<InputIndexSize>                         anchoring_period)
{
	const uint64_t n = num_positions, period = anchoring_period, blocks = (n + period - 1) / period;
	const uint64_t longest = ~(uint64_t) 0 >> (64 - 8 * RunLengthSize);
	for (uint64_t a = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; a < blocks; a += (uint64_t) gridDim.x * blockDim.x) {
		const uint64_t end = (a + 1) * period < n ? (a + 1) * period : n;
		uint64_t runs = 0, bytes = 0;
		for (uint64_t i = a * period; i < end; ) {
			uint64_t length; unsigned width;
			next_run<OutputIndexSize>(positions, i, end, longest, length, width);
			runs++;
			bytes += (length * (width / 8) + OutputIndexSize - 1) / OutputIndexSize * OutputIndexSize;
			i += length;
		}
		block_runs[a] = runs; block_bytes[a] = bytes;
	}
}

/// `run_offsets`, `byte_offsets`: exclusive prefix sums of count_runs' arrays
template <unsigned OutputIndexSize, unsigned InputIndexSize, unsigned RunLengthSize>
__global__ void __launch_bounds__(threads_per_block) write_runs(
	uint_t<RunLengthSize>*           __restrict__  scatter_position_run_lengths,
	unsigned char*                   __restrict__  scatter_position_run_individual_offset_sizes,
	uint_t<OutputIndexSize>*         __restrict__  scatter_position_run_baseline_values,
	uint_t<InputIndexSize>*          __restrict__  scatter_position_run_offsets_start_positions,
	unsigned char*                   __restrict__  scatter_position_offset_bytes,
	uint_t<InputIndexSize>*          __restrict__  scatter_position_anchors,
	const uint_t<OutputIndexSize>*   __restrict__  positions,
	const uint64_t*                  __restrict__  run_offsets,
	const uint64_t*                  __restrict__  byte_offsets,
	size_type_by_index_size<InputIndexSize>        num_positions,
	uint_t<InputIndexSize>                         anchoring_period)
{
	const uint64_t n = num_positions, period = anchoring_period, blocks = (n + period - 1) / period;
	const uint64_t longest = ~(uint64_t) 0 >> (64 - 8 * RunLengthSize);
	for (uint64_t a = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; a < blocks; a += (uint64_t) gridDim.x * blockDim.x) {
		const uint64_t end = (a + 1) * period < n ? (a + 1) * period : n;
		uint64_t run = run_offsets[a], at = byte_offsets[a];
		scatter_position_anchors[a] = (uint_t<InputIndexSize>) run;
		for (uint64_t i = a * period; i < end; ) {
			uint64_t length; unsigned width;
			next_run<OutputIndexSize>(positions, i, end, longest, length, width);
			const uint64_t base = positions[i];
			scatter_position_run_lengths[run] = (uint_t<RunLengthSize>) length;
			scatter_position_run_individual_offset_sizes[run] = (unsigned char) width;
			scatter_position_run_baseline_values[run] = (uint_t<OutputIndexSize>) base;
			scatter_position_run_offsets_start_positions[run] = (uint_t<InputIndexSize>) (at / OutputIndexSize);
			const unsigned bytes_each = width / 8;
			for (uint64_t t = 0; t < length; t++) {
				const uint64_t offset = (uint64_t) positions[i + t] - base;
				for (unsigned b = 0; b < bytes_each; b++) { scatter_position_offset_bytes[at + t * bytes_each + b] = (unsigned char) (offset >> (8 * b)); }
			}
			const uint64_t used = length * bytes_each, padded = (used + OutputIndexSize - 1) / OutputIndexSize * OutputIndexSize;
			for (uint64_t b = used; b < padded; b++) { scatter_position_offset_bytes[at + b] = 0; }
			at += padded; run++; i += length;
		}
		if (a + 1 == blocks) { for (uint64_t b = at; b % 8 != 0; b++) { scatter_position_offset_bytes[b] = 0; } }   // the pool ends on a multiple of 8
	}
}

} // namespace compressed_indices
} // namespace compression
} // namespace kernels
} // namespace cuda
