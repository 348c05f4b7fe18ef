// Host side of giddy/on_device/ordered_tiles.hpp: where a launch's ticket counters live and what they will read when it starts.
//
// Launches in one stream run one after the other, so ONE set of counters per (device, stream) serves them all: the counters only
// ever count up, and since the number of tickets a launch takes is known when it is enqueued (takes_of_launch), so is the value
// each counter has when the NEXT launch of the stream starts — it travels to the kernel as an argument.  No memset, no reset, no
// allocation on the launch path after a stream's first launch.  Counter sets come from arenas that are never freed (a process
// uses a handful of streams).
//
// A launch that is being captured into a graph would be replayed with the counter values of capture time: capturing launches
// get no tickets, and the wrappers then launch the kernel with the static tile order.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <unordered_map>

#include "giddy/api/types.hpp"
#include "giddy/on_device/ordered_tiles.hpp"

namespace cuda {
namespace ordered_tiles {

namespace detail {

struct registry {
	std::mutex mutex;
	struct key { int device; hipStream_t stream; bool operator==(const key& o) const { return device == o.device && stream == o.stream; } };
	struct hash { size_t operator()(const key& k) const { return std::hash<const void*>()(k.stream) * 31u + (size_t) k.device; } };
	struct entry { tickets* block; bases next_base; };
	std::unordered_map<key, entry, hash> entries;
	std::unordered_map<int, std::pair<tickets*, size_t>> arenas;   // per device: the current arena and how many counter sets of it are taken
	enum : size_t { sets_per_arena = 16 };

	static registry& instance() { static registry r; return r; }
};

} // namespace detail

/// GIDDY_ORDERED_TILES=0 in the environment turns ticketed tile order off (the kernels then map tiles to workgroups statically)
inline bool enabled()
{
	static const bool on = [] { const char* v = std::getenv("GIDDY_ORDERED_TILES"); return !(v && v[0] == '0'); }();
	return on;
}

/// what a ticketed kernel is given: the stream's counters and the values they have when this launch starts
struct for_launch { tickets* block; bases base; };

/// Tickets for the next launch on `stream` (of the current device): `grid` workgroups over `virtual_blocks` tiles, `classes` counters.
/// Returns false when there are none to be had (disabled, capturing, out of memory) — then: the static tile order.  The caller MUST
/// enqueue the launch if this returns true: the stream's counters are booked for it.
inline bool take(stream::id_t stream, unsigned long long grid, unsigned long long virtual_blocks, unsigned classes, for_launch& out)
{
	if (!enabled() || grid == 0) { return false; }
	int device = 0;
	if (hipGetDevice(&device) != hipSuccess) { return false; }
	hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
	if (hipStreamIsCapturing(stream, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) { (void) hipGetLastError(); return false; }
	auto& r = detail::registry::instance();
	std::lock_guard<std::mutex> lock(r.mutex);
	auto found = r.entries.find({ device, stream });
	if (found == r.entries.end()) {
		auto& arena = r.arenas[device];
		if (arena.first == nullptr || arena.second == detail::registry::sets_per_arena) {
			void* p = nullptr;
			if (hipMalloc(&p, detail::registry::sets_per_arena * sizeof(tickets)) != hipSuccess) { (void) hipGetLastError(); return false; }
			if (hipMemset(p, 0, detail::registry::sets_per_arena * sizeof(tickets)) != hipSuccess) { (void) hipGetLastError(); (void) hipFree(p); return false; }
			arena = { static_cast<tickets*>(p), 0 };
		}
		found = r.entries.emplace(detail::registry::key { device, stream }, detail::registry::entry { arena.first + arena.second++, bases { } }).first;
	}
	out.block = found->second.block;
	out.base = found->second.next_base;
	for (unsigned c = 0; c < classes; c++) { found->second.next_base.of_class[c] += takes_of_launch(grid, virtual_blocks, c, classes); }
	return true;
}

} // namespace ordered_tiles
} // namespace cuda
