// Device-side writer of incidence bitmaps: one bitmap per distinct value (at most 256), bit j of word w of bitmap b set where
// element 32 * w + j has dictionary key b; every bitmap padded to a multiple of BitmapAlignmentInInts words.
// Format: libgiddy_amd/formats.py::incidence_bitmaps (reference: incidence_bitmaps.cuh:63-131).  Input: the column's dictionary
// KEYS, one byte each (compression::dictionary::encode writes them; without a dictionary the keys are the values).
// A wavefront takes 64 consecutive elements = two words of every bitmap: one 64-lane ballot per bitmap.
#pragma once
#include "giddy/kernels/compression/common.hpp"

namespace cuda {
namespace kernels {
namespace compression {
namespace incidence_bitmaps {

template <unsigned IndexSize, unsigned BitmapAlignmentInInts>
__global__ void __launch_bounds__(threads_per_block) compress(
	standard_bit_container_t*  __restrict__  incidence_bitmaps,
	const uint8_t*             __restrict__  keys,
	size_type_by_index_size<IndexSize>       bitmap_length,       // in 32-bit words: the column has 32 * bitmap_length elements
	native_word_t                            num_bitmaps)
{
	const uint64_t words = bitmap_length, n = 32 * words;
	const uint64_t aligned = (words + BitmapAlignmentInInts - 1) / BitmapAlignmentInInts * BitmapAlignmentInInts;
	const uint64_t pairs = (aligned + 1) / 2;   // a wavefront's 64 elements are a pair of words
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) blockIdx.x * wave64::per_block() + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) gridDim.x * wave64::per_block();
	for (uint64_t pair = wave; pair < pairs; pair += num_waves) {
		const uint64_t element = 64 * pair + lane;
		const unsigned key = element < n ? keys[element] : ~0u;   // (padding: no bitmap has it)
		for (unsigned b = 0; b < num_bitmaps; b++) {
			const uint64_t incidence = wave64::ballot(key == b);
			if (lane == 0) {
				standard_bit_container_t* row = incidence_bitmaps + (uint64_t) b * aligned;
				row[2 * pair] = (standard_bit_container_t) incidence;
				if (2 * pair + 1 < aligned) { row[2 * pair + 1] = (standard_bit_container_t) (incidence >> 32); }
			}
		}
	}
}

} // namespace incidence_bitmaps
} // namespace compression
} // namespace kernels
} // namespace cuda
