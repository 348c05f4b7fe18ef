// Scratch memory for a wrapper's own use between the launches it enqueues.  The kernels of the reference's signatures have no
// scratch parameter; the wrappers that need per-tile state (chained tiles for long Delta segments, finer anchors for long
// discard-zero-bytes segments) get it here and fall back to their scratch-free paths when there is none to be had.
//
// Launches on one stream run in order, so ONE block per (device, stream) serves them all: it is kept between launches (grown when a
// column needs more — the old block is freed in stream order, behind the work that may still use it) and costs nothing on the
// launch path after a stream's first long column; taking stream-ordered memory from the runtime's pool for every launch instead
// (hipMallocAsync / hipFreeAsync) measured ~50 us per launch.  A stream that is being CAPTURED into a graph gets exactly that,
// though: allocation and release become nodes of the graph, and nothing of this cache is baked into it.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <mutex>
#include <unordered_map>

#include "giddy/api/types.hpp"

namespace cuda {
namespace stream_scratch {

/// what take() hands out; give it back with release() once the launches that use it are enqueued
struct block {
	void* memory = nullptr;
	bool transient = false;   // allocated for this use alone (a capturing stream): release() frees it in stream order
	explicit operator bool() const { return memory != nullptr; }
};

namespace detail {

struct cache {
	std::mutex mutex;
	struct key { int device; hipStream_t stream; bool operator==(const key& o) const { return device == o.device && stream == o.stream; } };
	struct hash { size_t operator()(const key& k) const { return std::hash<const void*>()(k.stream) * 31u + (size_t) k.device; } };
	struct entry { void* memory; size_t bytes; };
	std::unordered_map<key, entry, hash> entries;
	static cache& instance() { static cache c; return c; }
};

} // namespace detail

/// `bytes` of device memory usable by work enqueued on `stream` after this call (contents undefined); empty if there is none to be had
inline block take(stream::id_t stream, size_t bytes)
{
	block result;
	hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
	if (hipStreamIsCapturing(stream, &capturing) != hipSuccess) { (void) hipGetLastError(); return result; }
	if (capturing != hipStreamCaptureStatusNone) {
		if (hipMallocAsync(&result.memory, bytes, stream) != hipSuccess) { (void) hipGetLastError(); result.memory = nullptr; }
		result.transient = result.memory != nullptr;
		return result;
	}
	int device = 0;
	if (hipGetDevice(&device) != hipSuccess) { (void) hipGetLastError(); return result; }
	auto& c = detail::cache::instance();
	std::lock_guard<std::mutex> lock(c.mutex);
	auto& e = c.entries[{ device, stream }];
	if (e.bytes < bytes) {
		const size_t wanted = std::max<size_t>(bytes + bytes / 2, (size_t) 1 << 20);
		void* fresh = nullptr;
		if (hipMalloc(&fresh, wanted) != hipSuccess) { (void) hipGetLastError(); return result; }
		if (e.memory != nullptr) { (void) hipFreeAsync(e.memory, stream); }   // (behind the stream's earlier launches, which may still use it)
		e.memory = fresh; e.bytes = wanted;
	}
	result.memory = e.memory;
	return result;
}

/// ... zeroed (by a memset enqueued on `stream`)
inline block take_zeroed(stream::id_t stream, size_t bytes)
{
	block b = take(stream, bytes);
	if (b && hipMemsetAsync(b.memory, 0, bytes, stream) != hipSuccess) {
		(void) hipGetLastError();
		if (b.transient) { (void) hipFreeAsync(b.memory, stream); }
		b = block();
	}
	return b;
}

inline void release(stream::id_t stream, const block& b)
{
	if (b.memory != nullptr && b.transient) { (void) hipFreeAsync(b.memory, stream); }
}

} // namespace stream_scratch
} // namespace cuda
