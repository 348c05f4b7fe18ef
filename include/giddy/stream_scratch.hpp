// Scratch memory for a wrapper's own use between the launches it enqueues.  The kernels of the reference's signatures have no
// scratch parameter; the wrappers that need per-tile state (chained tiles for long Delta segments, finer anchors for long
// discard-zero-bytes segments) get it here and fall back to their scratch-free paths when there is none to be had.
//
// Launches on one stream run in order, so ONE block per (device, stream) serves them all: it is kept between launches and costs
// nothing on the launch path after a stream's first long column; taking stream-ordered memory from the runtime's pool for every
// launch instead (hipMallocAsync / hipFreeAsync) measured ~50 us per launch.  A block is LEASED from take() to release(): whoever asks
// for the stream's block while it is out — a wrapper's own scratch inside a pipeline chunk that already staged its slices in it — gets
// a transient block from the pool instead.  Growing waits for the stream (its earlier launches may still use the old block) — once in
// a stream's life for a given column size.  A stream that is being CAPTURED into a graph always gets transient blocks: allocation and
// release become nodes of the graph, and nothing of this cache is baked into it.
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
	bool transient = false;   // allocated for this use alone: release() frees it
	bool from_pool = false;   // ... from the runtime's stream-ordered pool (a capturing stream), freed in stream order
	unsigned slot = 0;
	explicit operator bool() const { return memory != nullptr; }
};

namespace detail {

struct cache {
	std::mutex mutex;
	struct key { int device; hipStream_t stream; unsigned slot; bool operator==(const key& o) const { return device == o.device && stream == o.stream && slot == o.slot; } };
	struct hash { size_t operator()(const key& k) const { return (std::hash<const void*>()(k.stream) * 31u + (size_t) k.device) * 7u + k.slot; } };
	struct entry { void* memory; size_t bytes; bool leased; };
	std::unordered_map<key, entry, hash> entries;
	static cache& instance() { static cache c; return c; }
};

} // namespace detail

/// `bytes` of device memory usable by work enqueued on `stream` after this call (contents undefined); empty if there is none to be had
/// `slot`: a stream has one kept block per slot (0: a launch's own scratch; 1: a pipeline call's per-column arrays, which live across the
/// launches of the call's chunks)
inline block take(stream::id_t stream, size_t bytes, unsigned slot = 0)
{
	block result;
	result.slot = slot;
	hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
	if (hipStreamIsCapturing(stream, &capturing) != hipSuccess) { (void) hipGetLastError(); return result; }
	if (capturing != hipStreamCaptureStatusNone) {
		if (hipMallocAsync(&result.memory, bytes, stream) != hipSuccess) { (void) hipGetLastError(); result.memory = nullptr; }
		result.transient = result.from_pool = result.memory != nullptr;
		return result;
	}
	int device = 0;
	if (hipGetDevice(&device) != hipSuccess) { (void) hipGetLastError(); return result; }
	auto& c = detail::cache::instance();
	std::lock_guard<std::mutex> lock(c.mutex);
	auto& e = c.entries[{ device, stream, slot }];
	if (e.leased) {
		// the stream's block is out: a block of this use's own.  (Not from the runtime's stream-ordered pool: with ROCm 7.2 memory from
		// hipMallocAsync held stale contents for the kernels that followed an asynchronous copy into it, late in a long process —
		// tests/test_gpu_sharding.py found it; plain allocations, freed once the stream has run past their use, are slower and safe.)
		if (hipMalloc(&result.memory, bytes) != hipSuccess) { (void) hipGetLastError(); result.memory = nullptr; }
		result.transient = result.memory != nullptr;
		return result;
	}
	if (e.bytes < bytes) {
		const size_t wanted = std::max<size_t>(bytes + bytes / 2, (size_t) 1 << 20);
		void* fresh = nullptr;
		if (hipMalloc(&fresh, wanted) != hipSuccess) { (void) hipGetLastError(); return result; }
		if (e.memory != nullptr) {
			(void) hipStreamSynchronize(stream);   // (the stream's earlier launches may still use the old block)
			(void) hipFree(e.memory);
		}
		e.memory = fresh; e.bytes = wanted;
	}
	e.leased = true;
	result.memory = e.memory;
	return result;
}

/// ... zeroed (by a memset enqueued on `stream`)
inline void release(stream::id_t stream, const block& b);

inline block take_zeroed(stream::id_t stream, size_t bytes)
{
	block b = take(stream, bytes);
	if (b && hipMemsetAsync(b.memory, 0, bytes, stream) != hipSuccess) {
		(void) hipGetLastError();
		release(stream, b);
		b = block();
	}
	return b;
}

inline void release(stream::id_t stream, const block& b)
{
	if (b.memory == nullptr) { return; }
	if (b.from_pool) { (void) hipFreeAsync(b.memory, stream); return; }
	if (b.transient) { (void) hipStreamSynchronize(stream); (void) hipFree(b.memory); return; }   // (a nested use of a stream's scratch: rare)
	int device = 0;
	if (hipGetDevice(&device) != hipSuccess) { (void) hipGetLastError(); return; }
	auto& c = detail::cache::instance();
	std::lock_guard<std::mutex> lock(c.mutex);
	auto found = c.entries.find({ device, stream, b.slot });
	if (found != c.entries.end() && found->second.memory == b.memory) { found->second.leased = false; }
}

/// take() ... release() as a scope: what a wrapper's enqueue_launch holds while it enqueues the launches that use the block
struct lease {
	stream::id_t stream;
	block taken;
	explicit lease(stream::id_t stream_) : stream(stream_) { }
	lease(const lease&) = delete;
	lease& operator=(const lease&) = delete;
	bool take(size_t bytes) { taken = stream_scratch::take(stream, bytes); return taken.memory != nullptr; }
	void* memory() const { return taken.memory; }
	~lease() { release(stream, taken); }
};

} // namespace stream_scratch
} // namespace cuda
