// Patches applied INSIDE a decoder: the reference decompresses a column and then
// launches data_layout::scatter over it to overwrite the few elements the scheme could not represent
// (README.md:54-60; src/kernels/data_layout/scatter.cuh:26-46) — a second launch and a scattered
// read-modify-write of HBM sectors.  When the patch positions are ASCENDING (any compressor emits
// them so; compressed-indices patches are ascending by construction) a wavefront that produces
// outputs in ascending order can merge them in registers just before its 16-byte stores:
//
//   * begin_seek() / finish_seek(): first patch at or after a position, found by the whole wavefront — a guess
//     from the patch density, checked 64 positions at a time (evenly spread patches: one load), else a
//     64-ary search narrowing the candidates 64-fold per load;
//   * cursor_t: a window of 64 patches (one per lane) with a wave-uniform read position; advancing
//     is two v_readlane, refilling is one coalesced load per 64 patches;
//   * apply(): overwrites the patched elements of a row of 64*E outputs held E per lane.
//
// `none` is the same interface doing nothing: the plain decoders are the patched ones instantiated
// with it, at no cost.
#pragma once
#include <hip/hip_runtime.h>

#include "giddy/common.hpp"
#include "giddy/on_device/wave64.hpp"

namespace cuda {
namespace sorted_patches {

/// Narrows a bracket [lo, hi] around the index of the first ascending position that is >= target to at most 64
/// candidates: every step probes 64 positions at once, a power-of-two stride apart (all arguments wave-uniform).
template <unsigned PositionSize>
__device__ __forceinline__ void narrow_lower_bound(
	const uint_t<PositionSize>* __restrict__ positions, uint64_t target, unsigned lane, uint64_t& lo, uint64_t& hi)
{
	while (hi - lo > 64) {
		const unsigned shift = (64 - (unsigned) __builtin_clzll(hi - lo - 1)) - 6;
		const uint64_t probe = lo + (((uint64_t) lane + 1) << shift) - 1;
		const bool below = probe < hi && (uint64_t) positions[probe] < target;
		lo += (uint64_t) __builtin_popcountll(wave64::ballot(below)) << shift;
		const uint64_t next_probe = lo + ((uint64_t) 1 << shift) - 1;   // the first probe that was not below, if it exists
		if (next_probe < hi) { hi = next_probe; }
	}
}

/// One thread on its own: index of the LAST patch at `position`, or `num` if there is none (scalar tails)
template <unsigned PositionSize>
__device__ __forceinline__ uint64_t find_last(const uint_t<PositionSize>* __restrict__ positions, uint64_t num, uint64_t position)
{
	uint64_t lo = 0, hi = num;   // first index whose position is > `position`
	while (lo < hi) {
		uint64_t mid = lo + (hi - lo) / 2;
		if ((uint64_t) positions[mid] <= position) { lo = mid + 1; } else { hi = mid; }
	}
	return (lo > 0 && (uint64_t) positions[lo - 1] == position) ? lo - 1 : num;
}

/// No patches: what the plain decoders are instantiated with.
struct none {
	enum : bool { enabled = false };
	__device__ __forceinline__ void begin_seek(uint64_t, unsigned) { }
	__device__ __forceinline__ void finish_seek(unsigned) { }
	__device__ __forceinline__ void seek(uint64_t, unsigned) { }
	struct handoff_t { };
	__device__ __forceinline__ void publish(handoff_t&, unsigned) const { }
	__device__ __forceinline__ void adopt(const handoff_t&, unsigned) { }
	template <unsigned E, typename Values> __device__ __forceinline__ void apply_to_row(uint64_t, unsigned, Values&) { }
	template <typename T> __device__ __forceinline__ T patched(uint64_t, T value) const { return value; }
};

template <unsigned PositionSize, typename Value>
struct cursor_t {
	enum : bool { enabled = true };
	enum : unsigned { guess_probes = 16 };
	using position_type = uint_t<PositionSize>;

	const position_type* __restrict__ positions;
	const Value* __restrict__ values;
	uint64_t num;
	float patches_per_output;                   // where to look first: patch index ~ output position * this
	unsigned guess_shift;                       // ... give or take 32 << guess_shift patches

	uint64_t lane_position; Value lane_value;   // the window: lane l holds patch window_start + l
	uint64_t window_start;                      // wave-uniform from here on
	unsigned index_in_window;
	uint64_t position;                          // the current patch; ~0 once the patches are exhausted
	Value value;

	/// `length`: the column's length in elements (only used to guess where a position's patches are)
	__device__ __forceinline__ cursor_t(const Value* values_, const position_type* positions_, uint64_t num_, uint64_t length)
	: positions(positions_), values(values_), num(num_), position(~(uint64_t) 0)
	{
		patches_per_output = length ? (float) num_ / (float) length : 0.0f;
		// evenly spread patches deviate from the guess by a few sqrt(num): reach (guess_probes / 2) << shift >= 2 * sqrt(num)
		const unsigned log_num = 64 - (unsigned) __builtin_clzll(num_ | 1);
		guess_shift = log_num / 2 > 10 ? log_num / 2 - 2 : 8;
	}

	__device__ __forceinline__ void read_current()
	{
		position = wave64::read_lane(lane_position, index_in_window);
		value = wave64::read_lane(lane_value, index_in_window);
	}
	__device__ __forceinline__ void fill_window(unsigned lane)
	{
		const uint64_t mine = window_start + lane;
		const bool have = mine < num;
		lane_position = have ? (uint64_t) positions[mine] : ~(uint64_t) 0;
		lane_value = have ? values[mine] : Value();
	}
	__device__ __forceinline__ void advance(unsigned lane)
	{
		if (++index_in_window == 64) { window_start += 64; fill_window(lane); index_in_window = 0; }
		read_current();
	}

	/// Position the cursor on the first patch at or after `first_output` (wave-uniform), in two halves so that the
	/// caller can put its own loads in between (CDNA4 returns loads in order; what is issued between the halves is
	/// not waited for by the first).  begin_seek() guesses where the patches of `first_output` are — evenly spread
	/// patches sit within a few sqrt(num) of first_output * num / length — and probes 16 positions `1 << guess_shift`
	/// apart around the guess; finish_seek() turns the answers into a bracket (falling back to the 64-ary search from
	/// whatever bracket the guess did establish) and loads its <= 64 candidates, which are at once the first window.
	uint64_t seek_target, seek_base, probed_position;

	__device__ __forceinline__ void begin_seek(uint64_t first_output, unsigned lane)
	{
		seek_target = first_output;
		if (num <= 64) { return; }
		uint64_t guess = (uint64_t) ((float) first_output * patches_per_output);
		if (guess > num) { guess = num; }
		const uint64_t reach = (uint64_t) (guess_probes / 2) << guess_shift;
		seek_base = guess > reach ? guess - reach : 0;
		// lane l < guess_probes asks about position base + (l << shift) - 1; with base == 0, lane 0 asks about
		// "position -1": below anything.  (Few probes on purpose: each is a cache line of its own.)
		const uint64_t probe = seek_base + ((uint64_t) lane << guess_shift) - 1;
		probed_position = (lane < guess_probes && probe < num) ? (uint64_t) positions[probe] : ~(uint64_t) 0;
	}
	__device__ __forceinline__ void finish_seek(unsigned lane)
	{
		uint64_t lo = 0, hi = num;   // the first patch at or after the target is in [lo, hi]
		if (num > 64) {
			const bool below = (lane == 0 && seek_base == 0) || probed_position < seek_target;
			const unsigned count = (unsigned) __builtin_popcountll(wave64::ballot(below));
			if (count == 0) { hi = seek_base - 1; }                       // even position base - 1 is not below
			else {
				lo = seek_base + ((uint64_t) (count - 1) << guess_shift);  // the last probe that was below sits just before this
				const uint64_t next_probe = seek_base + ((uint64_t) count << guess_shift) - 1;
				if (count < guess_probes && next_probe < hi) { hi = next_probe; }
			}
			narrow_lower_bound<PositionSize>(positions, seek_target, lane, lo, hi);
		}
		window_start = lo;
		fill_window(lane);
		// (positions from index hi on are >= the target, so counting over the whole window counts [lo, hi) only)
		index_in_window = (unsigned) __builtin_popcountll(wave64::ballot(lane_position < seek_target));
		if (index_in_window == 64) { window_start += 64; fill_window(lane); index_in_window = 0; }
		read_current();
	}
	__device__ __forceinline__ void seek(uint64_t first_output, unsigned lane) { begin_seek(first_output, lane); finish_seek(lane); }

	/// A sought cursor handed from one wavefront of a workgroup to the others through LDS (one seek per tile
	/// instead of one per wavefront): publish(), barrier, adopt().
	struct handoff_t { uint64_t window_start; unsigned index_in_window; uint64_t lane_position[64]; Value lane_value[64]; };

	__device__ __forceinline__ void publish(handoff_t& handoff, unsigned lane) const
	{
		handoff.lane_position[lane] = lane_position; handoff.lane_value[lane] = lane_value;
		if (lane == 0) { handoff.window_start = window_start; handoff.index_in_window = index_in_window; }
	}
	__device__ __forceinline__ void adopt(const handoff_t& handoff, unsigned lane)
	{
		lane_position = handoff.lane_position[lane]; lane_value = handoff.lane_value[lane];
		window_start = wave64::read_first_lane(handoff.window_start);
		index_in_window = wave64::read_first_lane(handoff.index_in_window);
		read_current();
	}

	/// Overwrite the patched elements among the 64*E outputs starting at `row_start`, lane l holding
	/// elements [l*E, l*E+E) in `row` (anything indexable).  Rows must come in ascending order after a
	/// seek(); patches before `row_start` (other wavefronts' rows) are skipped.
	template <unsigned E, typename Values>
	__device__ __forceinline__ void apply_to_row(uint64_t row_start, unsigned lane, Values& row)
	{
		while (position < row_start) { advance(lane); }
		while (position < row_start + 64 * E) {
			const unsigned offset = (unsigned) (position - row_start);
			if (lane == offset / E) {
				#pragma unroll
				for (unsigned j = 0; j < E; j++) { if (offset % E == j) { row[j] = value; } }
			}
			advance(lane);
		}
	}

	/// one element on its own thread (scalar tails and unaligned fall-backs): a binary search per element
	template <typename T>
	__device__ __forceinline__ T patched(uint64_t p, T decoded) const
	{
		const uint64_t at = find_last<PositionSize>(positions, num, p);
		return at < num ? (T) values[at] : decoded;
	}
};

} // namespace sorted_patches
} // namespace cuda
