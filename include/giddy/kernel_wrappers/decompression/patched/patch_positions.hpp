// Where a fused decoder's patches go: either the caller's plain ascending "patch_positions", or — when the
// argument map carries the compressed-indices arrays instead (the names of the reference's
// src/kernel_wrappers/data_layout/compressed_indices_scatter.cu:76-98, their index and run-length types those of the
// column's IndexSize) — those positions expanded into the caller's "patch_positions_scratch" (num_patches entries)
// by a small kernel enqueued just before the decoder on the same stream.  The expansion moves bytes proportional to
// the PATCHES; what the fusion removes is the second pass and the scattered read-modify-write over the COLUMN.
// PRECONDITIONS (not checked on the way): plain positions are ASCENDING (equal positions: the last one wins);
// compressed positions hold no run whose offset width is none of 8/16/32/64 bits (the scatter skips such runs; here
// the wavefront that meets one stops patching its chunk).  To find out, pass "unsupported_patch_count" (unsigned*,
// zeroed): the expansion counts the patches of such runs there.
#pragma once
#include "giddy/kernel_wrappers/data_layout/compressed_indices_positions.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace patched {

template <unsigned IndexSize>
inline const uint_t<IndexSize>* positions(stream::id_t stream, const cuda::kernel_t::arguments_type& arguments, uint_t<IndexSize> num_patches)
{
	using index_type = uint_t<IndexSize>;
	if (arguments.find("patch_positions") != arguments.end()) { return argument<const index_type*>(arguments, "patch_positions"); }
	auto scratch = argument<index_type*>(arguments, "patch_positions_scratch");
	scatter::compressed_indices::positions::enqueue_expansion<IndexSize, IndexSize>(stream, scratch, arguments, num_patches);
	return scratch;
}

} // namespace patched
} // namespace decompression
} // namespace kernels
} // namespace cuda
