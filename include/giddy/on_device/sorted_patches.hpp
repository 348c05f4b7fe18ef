// Patches applied INSIDE a decoder: the reference decompresses a column and then
// launches data_layout::scatter over it to overwrite the few elements the scheme could not represent
// (README.md:54-60; src/kernels/data_layout/scatter.cuh:26-46) — a second launch and a scattered
// read-modify-write of HBM sectors.  When the patch positions are ASCENDING (any compressor emits
// them so; compressed-indices patches are ascending by construction) a wavefront that produces
// outputs in ascending order can merge them in registers just before its 16-byte stores:
//
//   * begin_seek() / finish_seek(): first patch at or after a position, found by the whole wavefront — a guess
//     from the patch density, checked 16 positions at a time (evenly spread patches: one load), else a
//     64-ary search narrowing the candidates 64-fold per load;
//   * cursor_t: a window of 64 patches in wave-private LDS with a wave-uniform read position; advancing
//     is two LDS reads, refilling is one coalesced load per 64 patches;
//   * apply_to_row(): overwrites the patched elements of a row of 64*E outputs held E per lane.
//
// `none` is the same interface doing nothing: the plain decoders are the patched ones instantiated
// with it, at no cost.
#pragma once
#include <hip/hip_runtime.h>

#include "giddy/common.hpp"
#include "giddy/on_device/wave64.hpp"

#ifndef GIDDY_PATCH_WINDOW_PAGES
#define GIDDY_PATCH_WINDOW_PAGES 2   // ManyPatches builds: pages of 64 patches in the window of a wavefront that has a row area (one memory round trip refills them all).  1 | 2 | 4 at 5 % of patches: Dictionary 66.6 | 70.0 | 47.4 % of the HBM peak, at 50 %: 55.5 | 63.8 | 48.3; DZB-fixed, FOR, Model: +1 ... +6 with two; Delta 69.1 | 63.3 | 42.0 (its registers: one page there)
#endif
#ifndef GIDDY_PATCHES_TOGETHER_FROM
#define GIDDY_PATCHES_TOGETHER_FROM -1   // >= 0 (measurements): patches per row from which EVERY fused wrapper launches its ManyPatches build, instead of the table below
#endif

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
	enum : bool { enabled = false, many = false };
	__device__ __forceinline__ void begin_seek(uint64_t, unsigned) { }
	__device__ __forceinline__ void finish_seek(unsigned) { }
	__device__ __forceinline__ void seek(uint64_t, unsigned) { }
	template <unsigned E, typename Values> __device__ __forceinline__ void apply_to_row(uint64_t, unsigned, Values&) { }
	template <unsigned E, typename Values> __device__ __forceinline__ void apply_to_row(uint64_t, uint64_t, unsigned, Values&) { }
	template <typename T> __device__ __forceinline__ T patched(uint64_t, T value) const { return value; }
};

/// Patches per row of 64 lanes' elements, on the column's average, from which a fused decoder's ManyPatches build is the faster one — per decoder family,
/// where the two builds' rates cross (both builds forced, 2^28 elements: profiles/r05_patch_density.txt; round 4 had one constant, 6, and a dent under it).
/// The walk that patches one at a time costs the whole wavefront a trip around a loop of LDS reads per patch; the ManyPatches builds pay five LDS operations
/// per row whatever its patches (cursor_t::apply_to_row) — since they pass a row's patches through a row area in LDS the crossings are at 1 ... 3 patches per
/// row (with the search per lane they did before: 2 ... 4.5).
namespace together_from {
constexpr float delta = 1.5f, dictionary = 1.7f, discard_zero_bytes_fixed = 2.0f, frame_of_reference = 3.0f, model = 1.0f;
}

/// whether a column of `length` elements with `num_patches` patches, decoded in rows of 64 lanes x `elements_per_lane`, is one for the ManyPatches builds
inline bool many_patches_per_row(uint64_t num_patches, uint64_t length, unsigned elements_per_lane, float from)
{
	// (`from` was measured on 4- and 8-byte outputs — rows of 256 and 128 elements.  The rows of 2- and 1-byte outputs are 512 and 1024 elements long and their builds cross at about
	// twice as many patches per row — Dictionary <4,2,1> / <4,1,1>: 3.8 / 3.6 against 1.7, Delta <4,2,1>: 3.6 against 1.5, Model -> short 2.0 against 1.0; taken at `from` they lost
	// 6-13 points between 0.25 and 0.7 % of patches: profiles/r05_patch_density.txt, last table)
	const double threshold = GIDDY_PATCHES_TOGETHER_FROM >= 0 ? (double) GIDDY_PATCHES_TOGETHER_FROM : (double) from * (elements_per_lane >= 8 ? 2.0 : 1.0);
	return length != 0 && (double) num_patches * (double) (64 * elements_per_lane) >= threshold * (double) length;
}

/// Wave-private LDS for the cursors of a workgroup's wavefronts: a window of 64 patches each.  Declared __shared__
/// by the kernel (at most 16 wavefronts = 1024 threads per workgroup).
template <unsigned PositionSize, typename Value, unsigned MaxWaves = 16>
struct window_storage {
	enum : unsigned { max_waves_per_block = MaxWaves };
	uint_t<PositionSize> positions[max_waves_per_block][64];
	Value values[max_waves_per_block][64];
};

/// The `ManyPatches` builds' second wave-private area: ONE ROW of the output as LDS addresses — a value and a byte "a patch lies here" per element of a row
/// of 64 lanes' elements (16 bytes per lane: 1 KiB of values, 64 ... 1024 bytes of marks) — through which a row's patches reach the lanes that hold their
/// elements (cursor_t::apply_to_row).  `MaxWaves` 0: the other builds, which have no use for it.
/// ... and, with `WindowPages` > 1, a window of that many pages of 64 patches for the same wavefronts, in place of their one-page window of window_storage.
template <typename Value, unsigned MaxWaves, unsigned ElementsPerLane = 16 / sizeof(Value), unsigned PositionSize = 4, unsigned WindowPages = GIDDY_PATCH_WINDOW_PAGES>
struct row_storage {
	enum : unsigned { max_waves_per_block = MaxWaves, row_elements = 64 * ElementsPerLane, window_pages = WindowPages, wide_entries = WindowPages > 1 ? 64 * WindowPages : 1 };
	Value values[MaxWaves][row_elements];
	unsigned char marks[MaxWaves][row_elements];
	uint_t<PositionSize> window_positions[MaxWaves][wide_entries];
	Value window_values[MaxWaves][wide_entries];
};
template <typename Value, unsigned ElementsPerLane, unsigned PositionSize, unsigned WindowPages>
struct row_storage<Value, 0, ElementsPerLane, PositionSize, WindowPages> {
	enum : unsigned { max_waves_per_block = 0, row_elements = 0, window_pages = 1 };
};

/// `ManyPatches`: apply_to_row has every lane look for its own patches (columns with several patches to a row: the wrappers launch the kernels'
/// `ManyPatches` builds for them) instead of walking the patches one at a time
template <unsigned PositionSize, typename Value, bool ManyPatches = false, unsigned RowElementsPerLane = 16 / sizeof(Value), unsigned MaxWindowPages = (ManyPatches ? GIDDY_PATCH_WINDOW_PAGES : 1)>
struct cursor_t {
	enum : bool { enabled = true, many = ManyPatches };
	enum : unsigned { guess_probes = 16, max_window_pages = MaxWindowPages };
	using position_type = uint_t<PositionSize>;
	using storage_type = window_storage<PositionSize, Value>;
	static constexpr position_type no_patch = (position_type) ~(position_type) 0;   // (no column has an element at the all-ones index)

	// Everything here is wave-uniform (scalar registers) but `probed_position`; the window itself lives in
	// wave-private LDS, so that walking it costs LDS reads only.  (A window held in vector registers and refilled
	// by loads inside the row loops makes the compiler wait for ALL outstanding memory operations before every
	// access to it — CDNA4 counts them in order: the next rows' loads, the previous rows' stores — and costs the
	// streaming decoders a wavefront or two of occupancy per SIMD.)
	const position_type* __restrict__ positions;
	const Value* __restrict__ values;
	uint64_t num;
	float patches_per_output;                   // where to look first: patch index ~ output position * this
	unsigned guess_shift;                       // ... give or take (guess_probes / 2) << guess_shift patches
	position_type* window_positions;            // LDS: entry l is patch window_start + l (no_patch past the last one)
	Value* window_values;
	uint64_t window_start;
	unsigned index_in_window;
	unsigned window_entries = 64;               // 64 x the pages of this wavefront's window
	uint64_t position;                          // the current patch; ~0 once the patches are exhausted
	Value value;
	uint64_t seek_target, seek_base, probed_position;
	Value* row_values = nullptr;                // LDS (ManyPatches): this wavefront's row_storage
	unsigned char* row_marks = nullptr;

	/// `length`: the column's length in elements (only used to guess where a position's patches are)
	__device__ __forceinline__ cursor_t(const Value* values_, const position_type* positions_, uint64_t num_, uint64_t length, storage_type& storage)
	: cursor_t(values_, positions_, num_, length,
		storage.positions[wave64::index_in_block() % storage_type::max_waves_per_block],
		storage.values[wave64::index_in_block() % storage_type::max_waves_per_block]) { }

	/// ... and the ManyPatches builds' row area
	template <unsigned MaxWaves, unsigned ElementsPerLane, unsigned WindowPages>
	__device__ __forceinline__ cursor_t(const Value* values_, const position_type* positions_, uint64_t num_, uint64_t length, storage_type& storage, row_storage<Value, MaxWaves, ElementsPerLane, PositionSize, WindowPages>& rows)
	: cursor_t(values_, positions_, num_, length, storage)
	{
		with_rows(rows);
	}
	/// (for the callers that place the windows themselves)
	template <unsigned MaxWaves, unsigned ElementsPerLane, unsigned WindowPages>
	__device__ __forceinline__ cursor_t& with_rows(row_storage<Value, MaxWaves, ElementsPerLane, PositionSize, WindowPages>& rows)
	{
		if constexpr (MaxWaves != 0) {
			static_assert(ElementsPerLane >= RowElementsPerLane, "the row area holds the rows this cursor takes through it");
			// (a workgroup's first `MaxWaves` wavefronts have a row area — the wrappers launch workgroups of four; the wavefronts of a larger workgroup that a caller
			// of the bare kernel may launch find their patches by the search below)
			const unsigned w = wave64::index_in_block(), lane = wave64::lane_id();
			if (w < MaxWaves) {
				row_values = rows.values[w];
				row_marks = rows.marks[w];
				constexpr unsigned row_elements = row_storage<Value, MaxWaves, ElementsPerLane, PositionSize, WindowPages>::row_elements;
				if constexpr (WindowPages > 1) {
					static_assert(WindowPages <= max_window_pages, "fill_window() loads at most max_window_pages pages");
					window_positions = rows.window_positions[w];
					window_values = rows.window_values[w];
					window_entries = 64 * WindowPages;
				}
				for (unsigned i = lane * 4; i < row_elements; i += 256) { *reinterpret_cast<uint32_t*>(row_marks + i) = 0; }   // no patch anywhere, to begin with
				wave64::lds_fence();
			}
		}
		return *this;
	}

	/// ... with this wavefront's window (64 positions, 64 values of LDS) placed by the caller
	__device__ __forceinline__ cursor_t(const Value* values_, const position_type* positions_, uint64_t num_, uint64_t length,
		position_type* window_positions_, Value* window_values_)
	: positions(positions_), values(values_), num(num_), window_positions(window_positions_), window_values(window_values_), position(~(uint64_t) 0)
	{
		patches_per_output = length ? (float) num_ / (float) length : 0.0f;
		// evenly spread patches deviate from the guess by a few sqrt(num): reach (guess_probes / 2) << shift >= 2 * sqrt(num)
		const unsigned log_num = 64 - (unsigned) __builtin_clzll(num_ | 1);
		guess_shift = log_num / 2 > 10 ? log_num / 2 - 2 : 8;
	}

	__device__ __forceinline__ void read_current()
	{
		const position_type p = wave64::read_first_lane(window_positions[index_in_window]);
		position = p == no_patch ? ~(uint64_t) 0 : (uint64_t) p;
		value = wave64::read_first_lane(window_values[index_in_window]);
	}
	__device__ __forceinline__ void fill_window(unsigned lane)
	{
		// (Tried in round 4: the NEXT window's entries requested here, into registers, so that a column with a patch in every few elements does not wait
		// a memory round trip per window — loads in flight across the row loops make every wait the compiler places a wait for everything, as the note
		// above says: Delta with 0.1 % patches 72.6 -> 57.9 % of the HBM peak, nothing gained at 20-50 %.  Not kept.  Round 5: a window of TWO or FOUR pages of 64 patches,
		// refilled by one round trip — at 1-20 % of patches two pages +-1-2 points, four -15 ... -20 (their LDS costs the occupancy): the refills are not what the dense columns wait for.)
		// (a window of several pages: all its loads go out before any is waited for — one round trip)
		position_type p[max_window_pages];
		Value v[max_window_pages];
		#pragma unroll
		for (unsigned k = 0; k < max_window_pages; k++) {
			if (k == 0 || 64 * k < window_entries) {
				const uint64_t mine = window_start + 64 * k + lane;
				const bool have = mine < num;
				p[k] = have ? positions[mine] : no_patch;
				v[k] = have ? values[mine] : Value();
			}
		}
		wave64::lds_fence();   // (earlier reads of the window are done)
		#pragma unroll
		for (unsigned k = 0; k < max_window_pages; k++) {
			if (k == 0 || 64 * k < window_entries) { window_positions[64 * k + lane] = p[k]; window_values[64 * k + lane] = v[k]; }
		}
		wave64::lds_fence();
	}
	__device__ __forceinline__ void advance(unsigned lane)
	{
		if (++index_in_window == window_entries) { window_start += window_entries; index_in_window = 0; fill_window(lane); }
		read_current();
	}

	/// Position the cursor on the first patch at or after `first_output` (wave-uniform), in two halves so that the
	/// caller can put its own loads in between (CDNA4 returns loads in order; what is issued between the halves is
	/// not waited for by the first).  begin_seek() guesses where the patches of `first_output` are — evenly spread
	/// patches sit within a few sqrt(num) of first_output * num / length — and probes 16 positions `1 << guess_shift`
	/// apart around the guess; finish_seek() turns the answers into a bracket (falling back to the 64-ary search from
	/// whatever bracket the guess did establish) and loads its <= 64 candidates, which are at once the first window.
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
		const position_type mine = window_positions[lane];
		const bool before = mine != no_patch && (uint64_t) mine < seek_target;
		index_in_window = (unsigned) __builtin_popcountll(wave64::ballot(before));
		if (index_in_window == 64) { index_in_window = 63; advance(lane); }
		else { read_current(); }
	}
	__device__ __forceinline__ void seek(uint64_t first_output, unsigned lane) { begin_seek(first_output, lane); finish_seek(lane); }

	/// Overwrite the patched elements among the 64*E outputs starting at `row_start`, lane l holding
	/// elements [l*E, l*E+E) in `row` (anything indexable).  Rows must come in ascending order after a
	/// seek(); patches before `row_start` (other wavefronts' rows) are skipped.
	template <unsigned E, typename Values>
	__device__ __forceinline__ void apply_to_row(uint64_t row_start, unsigned lane, Values& row)
	{
		apply_to_row<E>(row_start, row_start + 64 * E, lane, row);
	}
	/// ... a row cut short at `row_end` (the end of an anchored segment): later patches stay for the rows to come
	template <unsigned E, typename Values>
	__device__ __forceinline__ void apply_to_row(uint64_t row_start, uint64_t row_end, unsigned lane, Values& row)
	{
		while (position < row_start) { advance(lane); }
		// Round 4: MANY patches in a row.  One at a time (below) a patch is a trip around a loop of LDS reads for the whole wavefront: fine for the one patch
		// in four rows of a 0.1 % column, 13 trips per row at 5 % (Dictionary: 31 % of the HBM peak), 128 at 50 % (6 %).  Where the column has `together_from`
		// patches or more to a row ON AVERAGE (decided once, by the wrapper, from the column's counts — and as a build of its own: finding out row by row cost the
		// 1 % columns 7 points, and the registers of this path in the same kernel cost the 0.1 % ones 2-5) a row's patches are taken together.
		if constexpr (ManyPatches) {
			// Round 5, second form: the row as LDS addresses.  The window's entry l belongs to lane l; the lanes whose entries lie in this row put their value and a mark at the
			// element's place in the wavefront's row area, every lane then reads the marks of its own E elements (one LDS read) and the values behind the set ones, and the
			// owners take their marks back: five LDS operations per row whatever the number of patches, no loop whose trip count differs between lanes.  (The first form: every
			// lane a binary search over the row's entries in the window — up to 6 dependent LDS reads — and then a step per patch of its own.)
			if constexpr (E <= RowElementsPerLane) {
				if (row_values != nullptr) {
					while (position < row_end) {
						const unsigned my_slot = (index_in_window & ~63u) + lane;   // (the page the current patch is in: a row's patches beyond it come a turn later)
						const position_type in_my_slot = window_positions[my_slot];
						const bool has_successor = my_slot + 1 < window_entries;
						const position_type in_the_next_slot = window_positions[has_successor ? my_slot + 1 : my_slot];
						const bool in_row = my_slot >= index_in_window && in_my_slot != no_patch && (uint64_t) in_my_slot < row_end;
						const unsigned count = (unsigned) __builtin_popcountll(wave64::ballot(in_row));   // (>= 1: the current patch is one of them)
						// (equal positions: the last wins — an entry whose successor in the window has the same position stands back; a successor in the NEXT window comes a turn later)
						const bool mine_to_place = in_row && !(has_successor && in_the_next_slot == in_my_slot);
						const unsigned slot = (unsigned) ((uint64_t) in_my_slot - row_start);
						if (mine_to_place) { row_values[slot] = window_values[my_slot]; row_marks[slot] = 1; }
						wave64::lds_fence();
						if constexpr (E % 4 == 0) {
							#pragma unroll
							for (unsigned q = 0; q < E / 4; q++) {
								const uint32_t marks = reinterpret_cast<const uint32_t*>(row_marks)[lane * (E / 4) + q];
								if (marks != 0) {
									#pragma unroll
									for (unsigned j = 0; j < 4; j++) { if ((marks >> (8 * j)) & 0xFFu) { row[4 * q + j] = row_values[lane * E + 4 * q + j]; } }
								}
							}
						}
						else {
							#pragma unroll
							for (unsigned j = 0; j < E; j++) { if (row_marks[lane * E + j]) { row[j] = row_values[lane * E + j]; } }
						}
						wave64::lds_fence();
						if (mine_to_place) { row_marks[slot] = 0; }
						index_in_window += count - 1;
						advance(lane);   // (to the entry after the row's last in this window: refills the window when that was its last)
					}
					return;
				}
			}
			// ... the first form: rows longer than the row area was made for (Delta's any-length rows of 64 x 4 eight-byte elements), wavefronts without one
			while (position < row_end) {
				const unsigned my_slot = (index_in_window & ~63u) + lane;
				const position_type in_my_slot = window_positions[my_slot];
				const bool in_row = my_slot >= index_in_window && in_my_slot != no_patch && (uint64_t) in_my_slot < row_end;
				const unsigned count = (unsigned) __builtin_popcountll(wave64::ballot(in_row));   // (>= 1: the current patch is one of them)
				const uint64_t my_first = row_start + (uint64_t) lane * E;
				const unsigned row_entries_end = index_in_window + count;
				unsigned low = index_in_window, high = row_entries_end;   // the first of the row's entries at or after this lane's first element
				while (low < high) {
					const unsigned middle = (low + high) / 2;
					if ((uint64_t) window_positions[middle] < my_first) { low = middle + 1; } else { high = middle; }
				}
				for (unsigned k = low; k < row_entries_end; k++) {
					const uint64_t at = (uint64_t) window_positions[k];
					if (at >= my_first + E) { break; }
					const unsigned offset = (unsigned) (at - my_first);
					const Value patch = window_values[k];
					#pragma unroll
					for (unsigned j = 0; j < E; j++) { if (offset == j) { row[j] = patch; } }
				}
				index_in_window = row_entries_end - 1;
				advance(lane);   // (to the entry after the row's last in this window: refills the window when that was its last)
			}
			return;
		}
		while (position < row_end) {
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
