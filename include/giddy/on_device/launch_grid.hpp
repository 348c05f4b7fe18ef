// Which workgroup of how many a kernel BODY is running as.  Every decoder body strides over its tiles / segments by
// (block index, number of blocks); for an ordinary launch those are the hardware's blockIdx.x / gridDim.x
// (`launch_grid::whole`: an empty object, the builtins — the bodies compile to what they were).  A BATCH launch
// (giddy/batch.hpp: many small columns decoded by ONE launch) hands each workgroup a part of the hardware grid —
// "block b of the k blocks of piece p" — and the same body runs on that (`launch_grid::part`): both values are
// wave-uniform and come from scalar loads, so they stay in scalar registers as the builtins do.
#pragma once
#include <hip/hip_runtime.h>

namespace cuda {
namespace launch_grid {

struct whole {
	__device__ __forceinline__ unsigned block() const { return blockIdx.x; }
	__device__ __forceinline__ unsigned blocks() const { return gridDim.x; }
};

struct part {
	unsigned block_, blocks_;
	__device__ __forceinline__ unsigned block() const { return block_; }
	__device__ __forceinline__ unsigned blocks() const { return blocks_; }
};

} // namespace launch_grid
} // namespace cuda
