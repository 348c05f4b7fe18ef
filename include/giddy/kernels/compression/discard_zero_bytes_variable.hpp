// Device-side compressor for variable-width discard-zero-bytes (null suppression): element i keeps its n_i = max(min size,
// significant bytes) low-order bytes; n_i - min size goes into a `bits`-wide field, floor(container bits / bits) fields per
// container, LSB first; position_anchors[s] = byte offset of element s * segment_length.
// Format: libgiddy_amd/formats.py::discard_zero_bytes_variable (reference: discard_zero_bytes_variable.cuh:23-33, :332-370).
// A THREAD takes one container's worth of consecutive elements (so a container is built in a register and written once);
// a tile = 256 containers.  Three launches (compression/common.hpp): byte counts per tile, offsets, write — the tile's bytes
// gathered in LDS and written out with 16-byte stores.
#pragma once
#include "giddy/kernels/compression/common.hpp"

namespace cuda {
namespace kernels {
namespace compression {
namespace discard_zero_bytes {
namespace variable_width {

/// bytes element `value` keeps: max(min size, significant bytes) — capped at the largest size a field can state (a value that needs
/// more is NOT representable with these parameters: it is stored truncated and counted by compress())
template <unsigned UncompressedSize>
__device__ __forceinline__ unsigned bytes_kept(uint_t<UncompressedSize> value, unsigned min_size, unsigned largest_size, bool& representable)
{
	const unsigned significant = value == 0 ? 0 : (64 - (unsigned) __builtin_clzll((unsigned long long) value) + 7) / 8;
	representable = significant <= largest_size;
	const unsigned wanted = significant < min_size ? min_size : significant;
	return wanted > largest_size ? largest_size : wanted;
}

__device__ __forceinline__ unsigned largest_size_of(unsigned min_size, unsigned bits, unsigned element_size)
{
	const uint64_t largest = (uint64_t) min_size + ((bits >= 32 ? (uint64_t) 1 << 32 : (uint64_t) 1 << bits) - 1);
	return (unsigned) (largest < element_size ? largest : element_size);
}

/// tile_totals[t] = bytes the elements of tile t (256 * per consecutive elements) keep
template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
__global__ void __launch_bounds__(threads_per_block) count_bytes(
	uint64_t*                        __restrict__  tile_totals,
	const uint_t<UncompressedSize>*  __restrict__  input,
	size_type_by_index_size<IndexSize>             length,
	native_word_t                                  min_represented_element_size,
	native_word_t                                  bits_per_element_size)
{
	__shared__ uint64_t table[waves_per_block + 1];
	const unsigned per = (8 * ElementSizesContainerSize) / bits_per_element_size;
	const unsigned largest_size = largest_size_of(min_represented_element_size, bits_per_element_size, UncompressedSize);
	const uint64_t n = length, tile = (uint64_t) threads_per_block * per, num_tiles = (n + tile - 1) / tile;
	for (uint64_t t = blockIdx.x; t < num_tiles; t += gridDim.x) {
		// (a tile's total does not care which thread adds which element: consecutive threads take consecutive elements — coalesced
		// loads, 9 x faster than a thread walking its own container's elements)
		const uint64_t first = t * tile + threadIdx.x;
		uint64_t mine = 0;
		bool representable;
		for (unsigned j = 0; j < per; j++) {
			const uint64_t p = first + (uint64_t) j * threads_per_block;
			if (p < n) { mine += bytes_kept<UncompressedSize>(input[p], min_represented_element_size, largest_size, representable); }
		}
		const uint64_t total = block_sum(mine, table);
		if (threadIdx.x == 0) { tile_totals[t] = total; }
	}
}

/// `tile_offsets`: the exclusive prefix sums of count_bytes' totals
template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
__global__ void __launch_bounds__(threads_per_block) compress(
	uint8_t*                                  __restrict__  compressed_data,
	uint_t<ElementSizesContainerSize>*        __restrict__  packed_element_sizes,
	uint_t<IndexSize>*                        __restrict__  position_anchors,
	const uint_t<UncompressedSize>*           __restrict__  input,
	const uint64_t*                           __restrict__  tile_offsets,
	size_type_by_index_size<IndexSize>                      segment_length,
	size_type_by_index_size<IndexSize>                      length,
	native_word_t                                           min_represented_element_size,
	native_word_t                                           bits_per_element_size,
	native_word_t*                            __restrict__  unrepresentable)
{
	using container_type = uint_t<ElementSizesContainerSize>;
	using chunk_t = typename memory_access::chunk<16>::type;
	enum : unsigned { staging_bytes = 40960 };
	// a tile's bytes are gathered here, laid out as they lie in global memory modulo 16, and leave as 16-byte stores
	// (element by element they would be one- to eight-byte stores, scattered: ~5 x slower)
	__shared__ __attribute__((aligned(16))) uint8_t staged[staging_bytes];
	__shared__ uint64_t table[waves_per_block + 1];
	const unsigned bits = bits_per_element_size, min_size = min_represented_element_size;
	const unsigned per = (8 * ElementSizesContainerSize) / bits;
	const unsigned largest_size = largest_size_of(min_size, bits, UncompressedSize);
	const uint64_t n = length, L = segment_length, tile = (uint64_t) threads_per_block * per, num_tiles = (n + tile - 1) / tile;
	const bool through_lds = tile * largest_size + 32 <= staging_bytes;   // (else: straight to global memory, byte by byte)
	unsigned bad = 0;
	for (uint64_t t = blockIdx.x; t < num_tiles; t += gridDim.x) {
		const uint64_t first = t * tile + (uint64_t) threadIdx.x * per;
		uint64_t mine = 0;
		bool representable;
		for (unsigned j = 0; j < per; j++) { if (first + j < n) { mine += bytes_kept<UncompressedSize>(input[first + j], min_size, largest_size, representable); } }
		uint64_t tile_total;
		const uint64_t tile_base = tile_offsets[t];
		const uint64_t before = block_exclusive_sum(mine, table, tile_total);
		uint64_t at = tile_base + before;
		const unsigned misalignment = (unsigned) ((reinterpret_cast<uintptr_t>(compressed_data) + tile_base) & 15);
		unsigned at_staged = misalignment + (unsigned) before;
		container_type container = 0;
		uint64_t until_anchor = first < n ? (L - first % L) % L : 0;   // elements until the next segment's first
		for (unsigned j = 0; j < per && first + j < n; j++) {
			const uint_t<UncompressedSize> value = input[first + j];
			const unsigned size = bytes_kept<UncompressedSize>(value, min_size, largest_size, representable);
			if (!representable) { bad++; }
			if (until_anchor == 0) { position_anchors[(first + j) / L] = (uint_t<IndexSize>) at; until_anchor = L; }
			until_anchor--;
			container |= (container_type) (size - min_size) << (j * bits);
			if (through_lds) {
				for (unsigned k = 0; k < size && k < 8; k++) { staged[at_staged + k] = (uint8_t) ((uint64_t) value >> (8 * k)); }
				at_staged += size;
			}
			else {
				for (unsigned k = 0; k < size && k < 8; k++) { compressed_data[at + k] = (uint8_t) ((uint64_t) value >> (8 * k)); }
			}
			at += size;
		}
		if (first < n) { packed_element_sizes[first / per] = container; }
		if (through_lds) {
			__syncthreads();
			uint8_t* const aligned_start = compressed_data + tile_base - misalignment;   // 16-byte aligned
			const unsigned valid_from = misalignment, valid_to = misalignment + (unsigned) tile_total, chunks = (valid_to + 15) / 16;
			for (unsigned c = threadIdx.x; c < chunks; c += blockDim.x) {
				const unsigned lo = 16 * c, hi = lo + 16;
				if (lo >= valid_from && hi <= valid_to) {
					*reinterpret_cast<chunk_t*>(aligned_start + lo) = *reinterpret_cast<const chunk_t*>(staged + lo);
				}
				else {   // the tile's first and last chunk are shared with its neighbours
					for (unsigned b = (lo < valid_from ? valid_from : lo); b < (hi < valid_to ? hi : valid_to); b++) { aligned_start[b] = staged[b]; }
				}
			}
			__syncthreads();   // (the next tile refills the staging area)
		}
	}
	if (unrepresentable != nullptr && bad != 0) { atomicAdd(unrepresentable, bad); }
}

} // namespace variable_width
} // namespace discard_zero_bytes
} // namespace compression
} // namespace kernels
} // namespace cuda
