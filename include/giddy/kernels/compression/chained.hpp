// One pass instead of three for the compressors whose output position depends on everything before it (compression/common.hpp
// describes the three: count per tile, offsets, write — the column read twice).  Tiles are CHAINED: a workgroup counts what its
// tile emits, publishes that total, adds up the published totals of the tiles before it — back to the nearest tile that already
// knows its own inclusive prefix ("decoupled look-back") — publishes its own prefix and writes.  The column is read once.
//
// A tile's state is one 8-byte word, {value << 2 | flag}, written and read with single relaxed agent-scope atomic accesses: it
// validates itself, no fence is needed, and nothing else is communicated between workgroups.  The states must be ZERO when the
// kernel starts (the wrappers enqueue a memset of the caller's `tile_totals` scratch).
//
// Progress: BOUNDED.  Every wait below counts its polls against a `patience` (a kernel argument; default_patience: seconds); a tile that runs out
// of it — or finds a tile before it marked `failed` — marks ITSELF failed, writes nothing and returns, so the failure runs down the chain
// at the speed of a look-back and the LAST tile stores `failed_base` (all ones) where the grand total goes: the caller reads that scalar
// anyway (it is the number of runs / bytes written) and sees that the column was not compressed.  Nothing hangs if the assumption
// below is ever violated (concurrent kernels from other streams, CU masks, a profiler that serialises dispatches ...).  `fault_tile`
// (a kernel argument, default none) makes one tile play dead — tests/test_gpu_compressors.py drives the whole path with it.
// The assumption: tile t waits only for tiles before it, and the hardware dispatches a launch's workgroups in index order on every XCD (the
// launch must have a workgroup per tile: a grid stride over more workgroups than are resident at once deadlocks — compressors.hpp) — the smallest unfinished tile's workgroup is therefore always resident and never waits
// for anything that has not been dispatched.  (The decoders' chained tiles for long Delta segments rest on the same fact.)
// Handing tiles out by an atomic ticket instead — so that a tile's predecessors really started before it, whichever XCD lags —
// was measured and is slower (run-position encoding, 2^28 elements: 0.67 ms against 0.50): the ticket is one more round trip on
// every workgroup's critical path.
//
// How far back: a probe reads 256 states (4 per lane of ONE wavefront) and takes about a microsecond; the nearest published prefix
// is as many tiles back as started during one look-back, so tiles must start slower than ~256 per microsecond chip-wide for the
// look-back to stay a probe or two long — 4096-element tiles of a 32-bit column at 4 TB/s start 250 per microsecond: the run
// finders take tiles of 8192.
#pragma once
#include "giddy/kernels/compression/common.hpp"

namespace cuda {
namespace kernels {
namespace compression {
namespace chained {

enum : uint64_t { nothing = 0, aggregate = 1, inclusive = 2, failed = 3, flag_mask = 3 };
constexpr uint64_t failed_base = ~(uint64_t) 0;       // what exclusive_base / sum_before return, and the grand total says, when the chain broke
constexpr uint64_t no_fault_tile = ~(uint64_t) 0;
enum : uint32_t { default_patience = 1u << 23 };      // polls (a poll: a sleep of ~0.3 us + a load): seconds before a tile gives up

__device__ __forceinline__ void publish(uint64_t* states, uint64_t tile, uint64_t value, uint64_t flag)
{
	__hip_atomic_store(states + tile, (value << 2) | flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ uint64_t peek(const uint64_t* states, uint64_t tile)
{
	return __hip_atomic_load(states + tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/// waits for `tile` to have said something; false: `polls` ran past `patience`
__device__ __forceinline__ bool wait_for(const uint64_t* states, uint64_t tile, uint32_t& polls, uint32_t patience)
{
	while ((peek(states, tile) & flag_mask) == nothing) {
		if (++polls > patience) { return false; }
		__builtin_amdgcn_s_sleep(8);
	}
	return true;
}

/// The sum of the totals of tiles [0, tile), into every lane — called by all 64 lanes of ONE wavefront, after the tile's own
/// total was published.  failed_base: a tile before this one failed, or did not speak within `patience` polls.
__device__ __forceinline__ uint64_t sum_before(const uint64_t* states, uint64_t tile, uint32_t patience = default_patience)
{
	constexpr unsigned per_lane = 4, window = 64 * per_lane;
	const unsigned lane = wave64::lane_id();
	uint64_t sum = 0, end = tile;   // tiles [0, end) are still to be added
	// First wait for the tile right before — ONE word, the same for all lanes: the tiles before it were dispatched earlier and have
	// said their part by then, nearly always.  (A thousand resident workgroups that each poll a window of sixteen cache lines of the
	// same few kilobytes swamp the memory channel those lie in: a poll took tens of microseconds.)
	uint32_t polls = 0;
	if (!wait_for(states, tile - 1, polls, patience)) { return failed_base; }
	while (end > 0) {
		// entry i of the window is tile end - 1 - i; before the column's first tile: an inclusive prefix of 0
		uint64_t words[per_lane];
		#pragma unroll
		for (unsigned j = 0; j < per_lane; j++) {
			const uint64_t i = lane * per_lane + j;
			words[j] = i < end ? peek(states, end - 1 - i) : (uint64_t) inclusive;
		}
		unsigned my_prefix = per_lane, my_nothing = per_lane;   // the nearest of mine that is a prefix / not there yet (a failed tile: as good as silent — and final)
		bool any_failed = false;
		#pragma unroll
		for (int j = per_lane - 1; j >= 0; j--) {
			if ((words[j] & flag_mask) == inclusive) { my_prefix = (unsigned) j; }
			if ((words[j] & flag_mask) == nothing || (words[j] & flag_mask) == failed) { my_nothing = (unsigned) j; }
			any_failed = any_failed || (words[j] & flag_mask) == failed;
		}
		const uint64_t with_failed = wave64::ballot(any_failed && my_nothing < per_lane);
		const uint64_t with_prefix = wave64::ballot(my_prefix < per_lane), with_nothing = wave64::ballot(my_nothing < per_lane);
		unsigned first_prefix = window, first_nothing = window;
		if (with_prefix != 0) { const unsigned l = (unsigned) __builtin_ctzll(with_prefix); first_prefix = l * per_lane + wave64::read_lane(my_prefix, l); }
		if (with_nothing != 0) { const unsigned l = (unsigned) __builtin_ctzll(with_nothing); first_nothing = l * per_lane + wave64::read_lane(my_nothing, l); }
		const bool reached_a_prefix = first_prefix < first_nothing;
		const unsigned usable = reached_a_prefix ? first_prefix + 1 : first_nothing;   // entries [0, usable) count
		if (!reached_a_prefix && with_failed != 0) {   // is the first entry that cannot be used a FAILED tile?  Then so is this one.
			const unsigned l = (unsigned) __builtin_ctzll(with_failed);
			uint64_t stopper = 0;
			#pragma unroll
			for (unsigned j = 0; j < per_lane; j++) { if (j == my_nothing) { stopper = words[j]; } }
			if (l * per_lane + wave64::read_lane(my_nothing, l) == first_nothing && (wave64::read_lane(stopper, l) & flag_mask) == failed) { return failed_base; }
		}
		if (usable == 0) {   // the nearest tile has not said anything yet: wait for that one word
			if (!wait_for(states, end - 1, polls, patience)) { return failed_base; }
			continue;
		}
		uint64_t mine = 0;
		#pragma unroll
		for (unsigned j = 0; j < per_lane; j++) { if (lane * per_lane + j < usable) { mine += words[j] >> 2; } }
		sum += wave64::read_last_lane(wave64::inclusive_sum(mine));
		if (reached_a_prefix) { break; }
		end -= usable;
	}
	return sum;
}

/// The tile's total said EARLY — as soon as it is known, before anything else the workgroup still has to do ahead of its look-back (the tiles
/// after this one wait for this word).  Any thread may call it; one does the work.  Then exclusive_base(..., already_announced = true).
__device__ __forceinline__ void announce(uint64_t* states, uint64_t tile, uint64_t tile_total)
{
	if (threadIdx.x == 0) { publish(states, tile, tile_total, tile == 0 ? inclusive : aggregate); }
}

/// What every chained tile does between counting and writing.  EVERY thread of the workgroup calls it (barriers inside) with the
/// tile's total; returns the sum of the totals of the tiles before `tile`.  `slot`: a uint64_t of LDS.  The last tile stores the
/// grand total to `grand_total` (if not null).
/// Returns failed_base — to every thread — when the chain broke before this tile (see "Progress" above): the tile is marked failed, the
/// last tile says so in `grand_total`, and the caller's workgroup must write nothing and return.
__device__ __forceinline__ uint64_t exclusive_base(uint64_t* states, uint64_t tile, uint64_t num_tiles, uint64_t tile_total, uint64_t* slot, uint64_t* grand_total, bool already_announced = false,
	uint32_t patience = default_patience)
{
#ifdef GIDDY_CHAINED_TIMING_WITHOUT_THE_CHAIN   // (measurements only: what the kernels cost without waiting for each other — wrong output)
	return 0;
#endif
	if (wave64::index_in_block() == 0) {
		const bool first_lane = wave64::lane_id() == 0;
		uint64_t base = 0;
		if (tile != 0) {
			if (first_lane && !already_announced) { publish(states, tile, tile_total, aggregate); }
			base = sum_before(states, tile, patience);
		}
		if (first_lane) {
			const bool broken = base == failed_base;
			publish(states, tile, broken ? 0 : base + tile_total, broken ? (uint64_t) failed : (uint64_t) inclusive);
			*slot = base;
			if (tile + 1 == num_tiles && grand_total != nullptr) { *grand_total = broken ? failed_base : base + tile_total; }
		}
	}
	__syncthreads();
	const uint64_t base = *slot;
	__syncthreads();   // (the slot is free for the workgroup's next tile)
	return base;
}

/// a tile gives up outside exclusive_base (a second chain of its own broke): marked failed in `states`; the last tile says so in `grand_total`
__device__ __forceinline__ void give_up(uint64_t* states, uint64_t tile, uint64_t num_tiles, uint64_t* grand_total)
{
	publish(states, tile, 0, failed);
	if (tile + 1 == num_tiles && grand_total != nullptr) { *grand_total = failed_base; }
}

} // namespace chained
} // namespace compression
} // namespace kernels
} // namespace cuda
