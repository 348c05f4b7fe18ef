// Finding a column's PATCHES on the device: the positions where what a lossy model / narrow encoding decodes to differs from
// the true column, and the true values there — ascending positions, as the scatter kernels and the fused decoders take them
// (format: libgiddy_amd/formats.py::patches; reference: scatter.cuh:13-31).  Three launches (compression/common.hpp).
#pragma once
#include "giddy/kernels/compression/common.hpp"

namespace cuda {
namespace kernels {
namespace compression {
namespace patches {

template <unsigned IndexSize, unsigned ElementSize>
__global__ void __launch_bounds__(threads_per_block) count_differences(
	uint64_t*                   __restrict__  tile_totals,
	const uint_t<ElementSize>*  __restrict__  decoded,
	const uint_t<ElementSize>*  __restrict__  truth,
	size_type_by_index_size<IndexSize>        length)
{
	__shared__ uint64_t table[waves_per_block + 1];
	const uint64_t n = length, num_tiles = (n + tile_elements - 1) / tile_elements;
	for (uint64_t t = blockIdx.x; t < num_tiles; t += gridDim.x) {
		const uint64_t first = t * tile_elements + (uint64_t) threadIdx.x * elements_per_thread;
		uint64_t mine = 0;
		for (unsigned j = 0; j < elements_per_thread && first + j < n; j++) { mine += decoded[first + j] != truth[first + j]; }
		const uint64_t total = block_sum(mine, table);
		if (threadIdx.x == 0) { tile_totals[t] = total; }
	}
}

template <unsigned IndexSize, unsigned ElementSize>
__global__ void __launch_bounds__(threads_per_block) write_differences(
	uint_t<IndexSize>*          __restrict__  patch_positions,
	uint_t<ElementSize>*        __restrict__  patch_values,
	const uint_t<ElementSize>*  __restrict__  decoded,
	const uint_t<ElementSize>*  __restrict__  truth,
	const uint64_t*             __restrict__  tile_offsets,
	size_type_by_index_size<IndexSize>        length,
	uint64_t                                  capacity)   // patches the two arrays have room for: patches beyond it are counted, not stored
{
	__shared__ uint64_t table[waves_per_block + 1];
	const uint64_t n = length, num_tiles = (n + tile_elements - 1) / tile_elements;
	for (uint64_t t = blockIdx.x; t < num_tiles; t += gridDim.x) {
		const uint64_t first = t * tile_elements + (uint64_t) threadIdx.x * elements_per_thread;
		uint64_t mine = 0;
		for (unsigned j = 0; j < elements_per_thread && first + j < n; j++) { mine += decoded[first + j] != truth[first + j]; }
		uint64_t tile_total;
		uint64_t k = tile_offsets[t] + block_exclusive_sum(mine, table, tile_total);
		if (mine != 0) {
			for (unsigned j = 0; j < elements_per_thread && first + j < n; j++) {
				const uint_t<ElementSize> value = truth[first + j];
				if (decoded[first + j] != value) { if (k < capacity) { patch_positions[k] = (uint_t<IndexSize>) (first + j); patch_values[k] = value; } k++; }
			}
		}
	}
}

} // namespace patches
} // namespace compression
} // namespace kernels
} // namespace cuda
