// Tiles handed out in the order workgroups ASK for them — a column's write front stays compact whatever the XCDs do.
//
// Why.  The dispatcher places workgroup b on XCD b % 8 and every XCD works through its own eighth of the grid at its own pace.
// On MI355X the paces differ (profiles/r03_dzbv_xcd_drift_2p30.txt: two of the eight XCDs ~10 % slower, a different two on another
// box), so with the static map "tile = blockIdx.x" the eight fronts drift apart — 5 MiB of output apart after 60 us of a launch,
// 130 MiB after 0.5 ms, 0.9 GiB after 2 ms — and the memory system, which serves a front of some tens of MiB at 81 % of the HBM
// peak, serves the scattered one at 65-70 %: the decoders lost 2.5 points per doubling of the column, and the launch ended with
// the slow XCDs working alone.  A launch boundary re-aligns the fronts (2^31 elements as sixteen launches: 78 % against 69 %) —
// this does the same without the boundaries: a workgroup takes a TICKET (one returning atomic add, agent scope) and decodes
// the tile of that number, so tiles begin in address order chip-wide and a faster XCD simply takes more of them.
//
// One counter word serves ~88 tickets per microsecond (MI355X_MICROARCH.md, row `dequeue`); decoders whose workgroups live for
// a few microseconds ask for more than that, so tickets come from `classes` interleaved counters: class c (of C) hands out the
// virtual blocks  8 * ((t / 8) * C + c) + t % 8  for t = 0, 1, 2 ... — eight in a row, then C - 1 groups of eight that belong to
// the other classes — and the workgroups of physical block b belong to class (b / 8) % C: every XCD serves every class, so the
// classes advance together.
//
// A counter is never reset: the host knows how many tickets of each class a launch takes (every workgroup takes tickets until
// one is past the end — or exactly one when the grid covers the tiles), so it gives every launch the values the counters will
// have when it starts (`bases`); launches in one stream are ordered, and the host side keeps one set of counters per stream
// (giddy/ordered_tiles.hpp).  Each counter has a 4 KiB page of its own: returning atomics on ONE 64-byte line — four counters
// and a completion count side by side were tried first — are served at ~80 per microsecond all together.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "giddy/on_device/wave64.hpp"

namespace cuda {
namespace ordered_tiles {

enum : unsigned { max_classes = 4, xcds = 8 };

struct tickets {
	struct alignas(4096) counter { unsigned long long taken; } of_class[max_classes];
};
struct bases { unsigned long long of_class[max_classes]; };

/// the t-th virtual block of class c, C classes
__host__ __device__ inline unsigned long long block_of_ticket(unsigned long long t, unsigned c, unsigned C)
{
	return xcds * ((t / xcds) * C + c) + t % xcds;
}
/// how many of the blocks 0 .. blocks-1 belong to class c
__host__ __device__ inline unsigned long long blocks_of_class(unsigned long long blocks, unsigned c, unsigned C)
{
	const unsigned long long period = (unsigned long long) xcds * C, whole = blocks / period, rest = blocks % period;
	const unsigned long long begin = (unsigned long long) xcds * c;
	return whole * xcds + (rest <= begin ? 0 : (rest - begin < xcds ? rest - begin : (unsigned long long) xcds));
}

/// A workgroup's source of tiles.  All threads of the workgroup call next() together; it returns the same virtual block
/// index to all of them, `none` when the launch's `virtual_blocks` are used up — a workgroup stops at its first `none`
/// (the host counts on exactly that, see takes_of_launch).
template <unsigned Classes>
struct queue {
	static_assert(Classes >= 1 && Classes <= max_classes, "1 to 4 ticket classes");
	enum : unsigned long long { none = ~0ull };
	tickets* block;
	bases base;
	unsigned long long virtual_blocks;
	unsigned long long* slot;   // one word of LDS through which the ticket reaches the other wavefronts

	__device__ __forceinline__ unsigned long long next()
	{
		const unsigned c = (blockIdx.x / xcds) % Classes;
		if (threadIdx.x == 0) {
			const unsigned long long t = __hip_atomic_fetch_add(&block->of_class[c].taken, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - base.of_class[c];
			*slot = block_of_ticket(t, c, Classes);
		}
		__syncthreads();
		const unsigned long long b = wave64::read_first_lane((uint64_t) *slot);
		__syncthreads();   // (the slot is rewritten by the next call)
		return b < virtual_blocks ? b : (unsigned long long) none;
	}
};

/// tickets of class c that a launch of `grid` workgroups over `virtual_blocks` tiles takes
__host__ __device__ inline unsigned long long takes_of_launch(unsigned long long grid, unsigned long long virtual_blocks, unsigned c, unsigned C)
{
	if (grid >= virtual_blocks) { return blocks_of_class(grid, c, C); }                          // one ticket per workgroup
	return blocks_of_class(virtual_blocks, c, C) + blocks_of_class(grid, c, C);                  // the tiles, and every workgroup's last ticket
}

} // namespace ordered_tiles
} // namespace cuda
