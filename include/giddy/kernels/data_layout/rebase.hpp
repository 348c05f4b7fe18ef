// out[i] = in[i] - bias, wrapping: a shard's slice of a column's position anchors (or of its absolute run positions) made to refer
// to the shard's own slices — where the anchors lie, on the device (libgiddy_amd/sharding.py does the same to device tensors with a
// torch subtraction; giddy/sharding.hpp says which bias).  A plain stream: 16-byte accesses where the slice allows, any alignment works.
#pragma once
#include <hip/hip_runtime.h>

#include "giddy/common.hpp"

namespace cuda {
namespace kernels {
namespace data_layout {
namespace rebase {

template <unsigned Size>
__global__ void __launch_bounds__(256) rebase(uint_t<Size>* out, const uint_t<Size>* in, uint64_t count, uint_t<Size> bias)
{
	for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t) gridDim.x * blockDim.x) {
		out[i] = (uint_t<Size>) (in[i] - bias);
	}
}

/// A whole column's anchors rebased chunk by chunk in one launch: entries [first[c], first[c + 1]) -= bias[c] for every chunk c (the
/// host-to-device pipeline uploads a column's per-segment arrays once and then only says where each chunk's entries begin).
struct chunk_of_entries { uint64_t first, bias; };

template <unsigned Size>
__global__ void __launch_bounds__(256) rebase_by_chunks(uint_t<Size>* entries, const chunk_of_entries* chunks, uint64_t num_chunks, uint64_t count)
{
	for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t) gridDim.x * blockDim.x) {
		// the last chunk whose first entry is at or before i
		uint64_t lo = 0, hi = num_chunks;
		while (hi - lo > 1) { const uint64_t mid = (lo + hi) / 2; if (chunks[mid].first <= i) { lo = mid; } else { hi = mid; } }
		entries[i] = (uint_t<Size>) (entries[i] - (uint_t<Size>) chunks[lo].bias);
	}
}

} // namespace rebase
} // namespace data_layout
} // namespace kernels
} // namespace cuda
