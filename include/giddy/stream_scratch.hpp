// Scratch memory for a wrapper's own use between the launches it enqueues.  The kernels of the reference's signatures have no
// scratch parameter (the wrappers are stateless, src/cuda/kernel_wrapper.cuh:35); the wrappers that need per-tile state (chained
// tiles for long Delta segments, finer anchors for long run / discard-zero-bytes segments, the host-to-device pipeline's slices)
// get it here and fall back to their scratch-free paths when there is none to be had.
//
// A process-wide POOL of plain hipMalloc blocks.  A block that is handed back remembers the (device, stream, host thread) that used it
// last, and an event is recorded behind that use: work enqueued on one stream by one thread runs in order, so that same stream and
// thread may have the block again at once; anybody else — another stream, the same handle in another thread — gets it once the event
// has completed (a query, never a wait), or a fresh block.  The host thread is part of the key because hipStreamPerThread — and the
// null stream under -fgpu-default-stream=per-thread — is ONE handle naming a different stream in every thread.  Nothing here
// synchronises: growing allocates a bigger block and keeps the smaller one for smaller requests (hipFree waits for the device, so
// blocks are only freed by trim(), which the caller calls when it can afford that).  Callers that take a new stream from a pool of
// streams for every column (torch.cuda.Stream() rotates through 32) therefore share a few blocks, not one per stream handle.
//
//   reserve(stream, bytes)   a block of at least `bytes` is there for `stream` afterwards: the launches that follow do not allocate
//   forget(stream)           call before destroying a stream: a later stream that happens to get the same handle inherits nothing
//   trim()                   frees every block that is not out on lease (waits for the work that used them); returns the bytes freed
//
// A stream that is being CAPTURED into a graph never gets pool blocks (a replay would use them while somebody else holds them): take()
// returns an empty block and the wrapper takes its scratch-free path inside graphs — see capture_policy below.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <mutex>
#include <thread>
#include <vector>

#include "giddy/api/types.hpp"

namespace cuda {
namespace stream_scratch {

/// what take() hands out; give it back with release() once the launches that use it are enqueued
struct block {
	void* memory = nullptr;
	size_t bytes = 0;
	bool from_runtime_pool = false;   // hipMallocAsync inside a capture (capture_policy::graph_allocation): released by hipFreeAsync
	explicit operator bool() const { return memory != nullptr; }
};

/// What a capturing stream gets.  `none`: nothing — the wrappers' scratch-free paths run inside graphs (the default: the graph holds no
/// pointer into this pool and no allocation nodes).  `graph_allocation`: hipMallocAsync / hipFreeAsync, which become nodes of the graph.
enum class capture_policy { none, graph_allocation };

namespace detail {

struct pool {
	struct entry {
		void* memory = nullptr;
		size_t bytes = 0;
		int device = 0;
		hipStream_t last_stream = nullptr;
		std::thread::id last_thread;
		hipEvent_t passed = nullptr;     // recorded behind the last use
		bool leased = false;
		bool bound = false;              // last_stream / last_thread may have it again without waiting
		bool recorded = false;           // `passed` says when anybody may
	};
	std::mutex mutex;
	std::vector<entry> entries;
	capture_policy in_capture = capture_policy::none;
	static pool& instance() { static pool p; return p; }
};

inline bool idle(pool::entry& e)
{
	if (!e.recorded) { return false; }
	const hipError_t status = hipEventQuery(e.passed);
	if (status == hipSuccess) { return true; }
	if (status != hipErrorNotReady) { (void) hipGetLastError(); }
	return false;
}

} // namespace detail

inline void set_capture_policy(capture_policy p)
{
	auto& pool = detail::pool::instance();
	std::lock_guard<std::mutex> lock(pool.mutex);
	pool.in_capture = p;
}

/// `bytes` of device memory usable by work enqueued on `stream` after this call (contents undefined); empty if there is none to be had
inline block take(stream::id_t stream, size_t bytes)
{
	block result;
	auto& pool = detail::pool::instance();
	hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
	if (hipStreamIsCapturing(stream, &capturing) != hipSuccess) { (void) hipGetLastError(); return result; }
	if (capturing != hipStreamCaptureStatusNone) {
		capture_policy policy;
		{ std::lock_guard<std::mutex> lock(pool.mutex); policy = pool.in_capture; }
		if (policy == capture_policy::graph_allocation && hipMallocAsync(&result.memory, bytes, stream) == hipSuccess) { result.bytes = bytes; result.from_runtime_pool = true; }
		else { (void) hipGetLastError(); result.memory = nullptr; }
		return result;
	}
	int device = 0;
	if (hipGetDevice(&device) != hipSuccess) { (void) hipGetLastError(); return result; }
	const auto thread = std::this_thread::get_id();
	std::lock_guard<std::mutex> lock(pool.mutex);
	// the smallest block that will do among those this stream (as this thread uses it) released last — ordered behind its earlier use by the stream itself —;
	// failing that, among the blocks whose last use has completed.  (One's own first: taking another stream's idle block while one's own would do leaves that
	// stream without its block the next time round, and its request — its own block taken, the taker's not yet idle — allocates: the pool grew by timing.)
	detail::pool::entry* best = nullptr;
	for (auto& e : pool.entries) {
		if (e.leased || e.device != device || e.bytes < bytes) { continue; }
		if (!(e.bound && e.last_stream == stream && e.last_thread == thread)) { continue; }
		if (best == nullptr || e.bytes < best->bytes) { best = &e; }
	}
	if (best == nullptr) {
		for (auto& e : pool.entries) {
			if (e.leased || e.device != device || e.bytes < bytes || !detail::idle(e)) { continue; }
			if (best == nullptr || e.bytes < best->bytes) { best = &e; }
		}
	}
	if (best == nullptr) {
		detail::pool::entry fresh;
		fresh.bytes = std::max<size_t>(bytes + bytes / 2, (size_t) 1 << 20);
		fresh.device = device;
		if (hipMalloc(&fresh.memory, fresh.bytes) != hipSuccess) { (void) hipGetLastError(); return result; }
		if (hipEventCreateWithFlags(&fresh.passed, hipEventDisableTiming) != hipSuccess) { (void) hipGetLastError(); fresh.passed = nullptr; }
		pool.entries.push_back(fresh);
		best = &pool.entries.back();
	}
	best->leased = true;
	best->bound = false;
	best->recorded = false;
	result.memory = best->memory;
	result.bytes = best->bytes;
	return result;
}

/// back to the pool: `stream` (as used by this host thread) may have it again at once
inline void release(stream::id_t stream, const block& b)
{
	if (b.memory == nullptr) { return; }
	if (b.from_runtime_pool) { (void) hipFreeAsync(b.memory, stream); return; }
	auto& pool = detail::pool::instance();
	std::lock_guard<std::mutex> lock(pool.mutex);
	for (auto& e : pool.entries) {
		if (e.memory != b.memory) { continue; }
		e.leased = false;
		e.bound = true;
		e.last_stream = stream;
		e.last_thread = std::this_thread::get_id();
		e.recorded = e.passed != nullptr && hipEventRecord(e.passed, stream) == hipSuccess;
		if (!e.recorded) { (void) hipGetLastError(); }
		return;
	}
}

/// ... zeroed (by a memset enqueued on `stream`)
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

/// a block of at least `bytes` is there for `stream` (as used by this thread) afterwards; false if the device has none
inline bool reserve(stream::id_t stream, size_t bytes)
{
	const block b = take(stream, bytes);
	release(stream, b);
	return (bool) b;
}

/// `stream` is about to be destroyed (or handed to another thread): its blocks lose their tie to the handle (the event recorded at
/// their release still says when they are idle)
inline void forget(stream::id_t stream)
{
	auto& pool = detail::pool::instance();
	const auto thread = std::this_thread::get_id();
	int device = 0;
	if (hipGetDevice(&device) != hipSuccess) { (void) hipGetLastError(); return; }
	std::lock_guard<std::mutex> lock(pool.mutex);
	for (auto& e : pool.entries) {
		if (!e.bound || e.leased || e.device != device || e.last_stream != stream || e.last_thread != thread) { continue; }
		e.bound = false;
	}
}

/// frees every block that is not out on lease — waiting for the work that used it —; returns the bytes given back to the device
inline size_t trim()
{
	auto& pool = detail::pool::instance();
	std::vector<detail::pool::entry> going;
	{
		std::lock_guard<std::mutex> lock(pool.mutex);
		auto kept = std::partition(pool.entries.begin(), pool.entries.end(), [](const detail::pool::entry& e) { return e.leased; });
		going.assign(kept, pool.entries.end());
		pool.entries.erase(kept, pool.entries.end());
	}
	size_t freed = 0;
	for (auto& e : going) {
		if (e.recorded) { (void) hipEventSynchronize(e.passed); }
		if (hipFree(e.memory) == hipSuccess) { freed += e.bytes; } else { (void) hipGetLastError(); }   // (hipFree waits for the device's pending work)
		if (e.passed != nullptr) { (void) hipEventDestroy(e.passed); }
	}
	return freed;
}

/// bytes the pool holds at the moment (leased or not)
inline size_t held_bytes()
{
	auto& pool = detail::pool::instance();
	std::lock_guard<std::mutex> lock(pool.mutex);
	size_t total = 0;
	for (const auto& e : pool.entries) { total += e.bytes; }
	return total;
}

/// take() ... release() as a scope: what a wrapper's launch holds while it enqueues the launches that use the block
struct lease {
	stream::id_t stream;
	block taken;
	explicit lease(stream::id_t stream_) : stream(stream_) { }
	lease(const lease&) = delete;
	lease& operator=(const lease&) = delete;
	bool take(size_t bytes) { release(stream, taken); taken = stream_scratch::take(stream, bytes); return taken.memory != nullptr; }
	bool take_zeroed(size_t bytes) { release(stream, taken); taken = stream_scratch::take_zeroed(stream, bytes); return taken.memory != nullptr; }
	void* memory() const { return taken.memory; }
	~lease() { release(stream, taken); }
};

} // namespace stream_scratch
} // namespace cuda
