// Scratch memory for a wrapper's own use between the launches it enqueues: stream-ordered (hipMallocAsync / hipFreeAsync from the
// device's default memory pool), so taking and releasing it costs no synchronisation and after a stream's first launches no
// allocation either — the pool hands the same block back.  The kernels of the reference's signatures have no scratch parameter;
// the wrappers that need per-tile state (chained tiles for long Delta segments, finer anchors for long discard-zero-bytes and
// run segments) get it here and fall back to their scratch-free paths when there is none to be had.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "giddy/api/types.hpp"

namespace cuda {
namespace stream_scratch {

/// `bytes` of device memory, zeroed, usable by work enqueued on `stream` after this call; nullptr if the runtime has none to give
inline void* take_zeroed(stream::id_t stream, size_t bytes)
{
	void* p = nullptr;
	if (hipMallocAsync(&p, bytes, stream) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
	if (hipMemsetAsync(p, 0, bytes, stream) != hipSuccess) { (void) hipGetLastError(); (void) hipFreeAsync(p, stream); return nullptr; }
	return p;
}

/// ... not zeroed
inline void* take(stream::id_t stream, size_t bytes)
{
	void* p = nullptr;
	if (hipMallocAsync(&p, bytes, stream) != hipSuccess) { (void) hipGetLastError(); return nullptr; }
	return p;
}

/// gives the memory back once the work enqueued on `stream` so far has run
inline void release(stream::id_t stream, void* p)
{
	if (p != nullptr) { (void) hipFreeAsync(p, stream); }
}

} // namespace stream_scratch
} // namespace cuda
