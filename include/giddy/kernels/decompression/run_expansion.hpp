// Shared machinery of the run-length (RLE) and run-position (RPE) decoders on CDNA4.
//
// Both schemes reduce to:  "given the start positions of the runs that begin inside a tile of the
// output, write run_data[rank] at every position, where rank = number of run starts at or before
// it".  The reference lets every thread re-walk the run arrays from the segment's first run
// (run_length_encoding.cuh:129-157, run_position_encoding.cuh:111-150: 1 + W/A loop iterations per
// element by its own cost model, at most two warps per block).  Here a WAVEFRONT owns a tile of up
// to 4096 outputs and uses a rank/select structure in LDS:
//
//   1. clear a 4096-bit tile bitmap (128 words, two per lane)                      — ds_write
//   2. mark: one LDS atomic-or per run start inside the tile                        — ds_or
//   3. rank: popcount per word + one 64-lane DPP prefix sum -> runs before each word — 6 VALU steps
//   4. expand: a lane owning E = 16/size consecutive outputs reads ONE {word, rank} pair (8-byte
//      LDS read), derives its E ranks with masked popcounts, looks the run values up in a
//      wave-private LDS copy of the tile's values (value_staging below) and issues one 16-byte
//      store, contiguous across the wave.
//
// No __syncthreads anywhere: a wavefront's LDS traffic is ordered by wavefront-scope fences, so
// wavefronts never wait for each other.  Work per output is constant, independent of run lengths.
#pragma once
#include <hip/hip_runtime.h>

#include "giddy/common.hpp"
#include "giddy/kernels/tuning.hpp"
#include "giddy/on_device/wave64.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace run_expansion {

// What the run decoders' tuning settled on (MI355X; DESIGN.md section 4 has each measurement).  No longer build switches:
//   * every tile's run values go through wave-private LDS, whatever the widths of values and descriptors (with write-through stores: staging wins for all);
//   * a lane's 2 or 4 look-ups of wide staged values go out together; rows of a tile begin on 512-byte boundaries of the output; a workgroup's chunks share
//     an aligned window of the output; chunks of many runs take their descriptors and values from LDS in groups; a batch of segments is a whole number of
//     64-byte blocks of output;
//   * tried and taken out: descriptors loaded around the L1 (-37 points), an anchor touched some thousand chunks ahead (-6...7 cold), 16 or 32 KiB of whole
//     segments per trip for 4- and 8-byte columns (-10), run values next to the descriptors in the dense-only builds, a residency cap for 8-byte descriptors.
enum : unsigned {
	segments_per_trip_at_least   = 2,      // segments shorter than a tile are taken at least this many at a time
	prefetch_from_element_size   = 2,      // element sizes from which the next tile's descriptors are requested ahead of the stores (1-byte columns: the registers are worth more)
	narrow_chunk_bytes           = 8192,   // output a wavefront of a 1-byte column takes per trip when segments are at least a tile long (two segments of 4096: +2-5 points for run lengths; 16384: mixed; not for relative run positions: -15)
	resident_blocks_cap          = 6,      // resident workgroups per CU for 32-bit values with 1-2-byte run descriptors (warm 5 is better, cold 6-8: the column that is decoded once decides)
};

enum : unsigned {
	words_per_lane      = 2,
	tile_words          = 64 * words_per_lane,     // 128
	tile_length         = 32 * tile_words,         // 4096 outputs per tile
	runs_per_lane       = 4,                       // run descriptors examined per lane per batch (6 / 8: +1...+3 / -4...-14 points)
	runs_per_batch      = 64 * runs_per_lane,
	max_waves_per_block = 4,                       // LDS is provisioned for this many wavefronts per workgroup (round 5: workgroups of 1 or 2 wavefronts, LDS and residency to match: 81 -> 72-73 % at 2^30, 73 -> 68 at 2^27)
};

/// Which chunk a wavefront starts with, chosen by where the OUTPUT lies.  A workgroup's wavefronts write `working_waves` consecutive
/// chunks; when those make up a power-of-two number of bytes — 4 x 16 KiB, a 64 KiB page, in configuration C4 — the grouping is
/// shifted so that every workgroup's chunks lie in ONE naturally aligned window of that size wherever the column begins (measured
/// with the static grouping, tools/alignment_probe.py: 85 % of the HBM peak on a 64 KiB boundary, 79 % displaced by 16 or 32 KiB —
/// the displacement a pool allocator or a shard boundary gives).  Global wavefront g takes chunks g - shift, g - shift + W, ...; the
/// `shift` wavefronts that would start below zero start one lap later, i.e. they take the column's last chunks.
__device__ __forceinline__ uint64_t first_chunk_by_address(const void* output, uint64_t chunk_bytes, unsigned working_waves, uint64_t wave, uint64_t num_waves)
{
	const uint64_t group_bytes = chunk_bytes * working_waves;
	if (chunk_bytes == 0 || (group_bytes & (group_bytes - 1)) != 0 || group_bytes > (1u << 20)) { return wave; }
	// (chunk_bytes and working_waves are powers of two as well: shifts and masks — two 64-bit divisions here cost the one- and two-byte
	// columns, whose wavefronts live for a 4 KiB tile, 5 to 15 points: profiles/r03_sweep_all_instantiations.txt against its successor)
	const uint64_t shift = ((reinterpret_cast<uintptr_t>(output) & (group_bytes - 1)) >> __builtin_ctzll(chunk_bytes)) & (working_waves - 1);
	return wave >= shift ? wave - shift : wave + num_waves - shift;
}

/// a run descriptor (length or position): read once
template <typename T>
__device__ __forceinline__ T load_descriptor(const T* p)
{
	return *p;   // (through the L1: a lane's four narrow loads of one line hit it — around it: -37 points)
}

struct tile_entry {
	uint32_t bits;         // bit j: a run starts at tile position 32*word + j   (position 0 is never marked)
	uint32_t rank_before;  // number of marked positions in all lower words
};

__device__ __forceinline__ void clear(tile_entry* table, unsigned lane)
{
	#pragma unroll
	for (unsigned i = 0; i < words_per_lane; i++) { table[words_per_lane * lane + i].bits = 0; }
	wave64::lds_fence();
}

__device__ __forceinline__ void mark(tile_entry* table, uint32_t position_in_tile)
{
	atomicOr(&table[position_in_tile >> 5].bits, 1u << (position_in_tile & 31));
}

/// fills rank_before; returns the number of run starts marked in the tile (wave-uniform)
__device__ __forceinline__ uint32_t rank(tile_entry* table, unsigned lane)
{
	wave64::lds_fence();
	uint32_t counts[words_per_lane], total = 0;
	#pragma unroll
	for (unsigned i = 0; i < words_per_lane; i++) { counts[i] = __builtin_popcount(table[words_per_lane * lane + i].bits); total += counts[i]; }
	uint32_t inclusive = wave64::inclusive_sum(total);
	uint32_t before = inclusive - total;
	#pragma unroll
	for (unsigned i = 0; i < words_per_lane; i++) { table[words_per_lane * lane + i].rank_before = before; before += counts[i]; }
	wave64::lds_fence();
	return wave64::read_last_lane(inclusive);
}

/// A tile's run values are staged into wave-private LDS first (one coalesced sweep) and looked up there:
///  * narrow elements (1-2 bytes) are looked up E = 8..16 times per lane — through the vector memory pipeline that
///    is 64 byte-sized gathers per instruction and it, not HBM, becomes the limit;
///  * wide elements (8 bytes) have only 128 outputs per row, so a tile is 32 rows whose gathers, one dependent
///    L2 round trip each, serialise (measured at 2^30 outputs, run values beyond the last-level cache: 49 %
///    of peak with gathers).
///  * 4-byte elements: with `nt` stores the look-ups through L1/L2 were 4 points faster than staged values when the run
///    descriptors (lengths or positions) are 1-2 bytes wide — 79.5 vs 75.9 % of peak for RLE <4,4,2> — and 12-15 points slower
///    when they are 4-8 bytes wide.  A tile expanded from staged values issues no load between its stores, so its stores can be
///    write-through (`staged_store_policy`), which the look-up path cannot afford; with that, staging wins for both (C4: RLE 84.7 ->
///    85.6 %, mean run 6: 68 -> 72 %) and is what every instantiation does now.
/// Tiles with more runs than fit fall back to gathers through L1/L2.
///
/// DENSE tiles (run_length_encoding.hpp / run_position_encoding.hpp): when the first batch of run descriptors does not reach the
/// end of the tile — mean runs below ~16 — the tile is CUT SHORT at what that batch covers (whole 128-byte pieces of output), so
/// that a tile never needs a second, dependent batch of descriptors, and its run values — at most a batch's worth — always come
/// from LDS: with runs of 2-4 elements the look-ups through L1/L2 miss (a tile's values are kilobytes, touched once), one
/// dependent HBM round trip PER ROW of 256 outputs — RLE<4,4,2> at mean run 4: 37 % of peak before.  `dense_only`: staging exists
/// for such tiles alone (the instantiations whose sparse tiles are faster through L1/L2).
template <unsigned ElementSize, unsigned RunDescriptorSize = 2>
struct value_staging {
	enum : unsigned { always = 1, dense_only = !always, enabled = 1,
		bytes = 1024 * words_per_lane, capacity = bytes / ElementSize,
		allocated = capacity + 16 / ElementSize,   // (the permuting expansions read up to 11 bytes past a lane's first run)
		// wide values are staged by the loop that loads the run descriptors (one dependent round trip less per tile: +1...4 points);
		// narrow ones by one sweep of their own inside expand() (element-sized loads next to every descriptor cost them 2-7 points)
		with_descriptors = always && ElementSize >= 4,
		dense_granule = 128 / ElementSize };   // a tile that is cut short ends on a multiple of this many outputs
};

/// A wavefront's chunk of `count` whole segments had better be a whole number of 64-byte blocks of output (round 4, run lengths in 4-byte elements at 2^28 outputs:
/// chunks of 40 segments of 100, 4 of 900 or 2 of 2000 — 14 400 or 16 000 bytes — 70 % of the HBM peak; of 13 segments of 300, 5 of 700 or 3 of 1100 — 15 600,
/// 14 000, 13 200 bytes — 65-66 %): `count` rounded down to the multiple that makes it so, where there is one (kernel and launch parameters agree; rounding UP to the first multiple, within
/// two tiles, gained 2 points at 1100 and 1500 and broke the relative positions' search over a chunk's anchors: not kept).
__host__ __device__ inline uint64_t whole_64_byte_blocks(uint64_t count, uint64_t segment_length, unsigned element_size)
{
	if (count <= 1) { return count; }
	uint64_t unit = 64, rest = (segment_length * element_size) % 64;
	while (rest != 0) { const uint64_t t = unit % rest; unit = rest; rest = t; }   // gcd(64, bytes per segment)
	const uint64_t multiple = 64 / unit;
	return count >= multiple ? count / multiple * multiple : count;
}

/// Groups of runs (run_length_encoding.hpp, run_position_encoding.hpp): for chunks of many runs the descriptors and values of up to
/// `group_runs` consecutive runs are brought into LDS by three coalesced 16-byte-per-lane loads, and the chunk's tiles are decoded
/// from there.  4-byte values with 1-2-byte descriptors (the value area of value_staging holds 512 of them; a group begins on a
/// multiple of 8 runs so that its loads are 16-byte aligned, and overlaps its predecessor by `overlap` runs: a tile that is cut at a
/// group's end hands over at most a 128-byte piece of output — 32 runs of one element at worst — before that end).
template <unsigned ElementSize, unsigned RunDescriptorSize, bool Allowed = true>
struct run_groups {
	enum : unsigned {
		enabled = Allowed && ElementSize == 4 && (RunDescriptorSize == 2 || RunDescriptorSize == 4),
		group_runs = 512,
		descriptors_per_lane = 16 / RunDescriptorSize, values_per_lane = 16 / ElementSize,
		descriptor_loads = (group_runs + 64 * descriptors_per_lane - 1) / (64 * descriptors_per_lane),   // 1 or 2 loads per lane of descriptors
		value_loads = (group_runs + 64 * values_per_lane - 1) / (64 * values_per_lane),                 // 2 loads per lane of values
		allocated = group_runs + 8, overlap = 48 };
	static_assert(!enabled || group_runs <= value_staging<ElementSize, RunDescriptorSize>::capacity, "a group's values fit the value staging area");
	typedef uint32_t sixteen_bytes __attribute__((ext_vector_type(4)));

	/// a group on its way from memory (registers), and in LDS
	struct in_flight { sixteen_bytes descriptors[descriptor_loads], values[value_loads]; };

	/// the group of runs [base, base + group_runs) requested (base: a multiple of 8; only 16-byte pieces that hold a run below `num_runs` are read)
	__device__ static __forceinline__ void request(in_flight& coming, const uint_t<RunDescriptorSize>* __restrict__ descriptors,
		const uint_t<ElementSize>* __restrict__ values, uint64_t base, uint64_t num_runs, unsigned lane)
	{
		#pragma unroll
		for (unsigned v = 0; v < descriptor_loads; v++) {
			const uint64_t run = base + (uint64_t) (v * 64 + lane) * descriptors_per_lane;
			if (run < num_runs && run < base + group_runs) { coming.descriptors[v] = memory_access::load_global<sixteen_bytes>(reinterpret_cast<uintptr_t>(descriptors + run)); }
		}
		#pragma unroll
		for (unsigned v = 0; v < value_loads; v++) {
			const uint64_t run = base + (uint64_t) (v * 64 + lane) * values_per_lane;
			if (run < num_runs && run < base + group_runs) { coming.values[v] = memory_access::load_global<sixteen_bytes>(reinterpret_cast<uintptr_t>(values + run)); }
		}
	}
	/// ... and put into LDS (entry i of either area: run base + i)
	__device__ static __forceinline__ void into_lds(const in_flight& coming, uint_t<RunDescriptorSize>* staged_descriptors, uint_t<ElementSize>* staged_values, unsigned lane)
	{
		#pragma unroll
		for (unsigned v = 0; v < descriptor_loads; v++) { reinterpret_cast<sixteen_bytes*>(staged_descriptors)[v * 64 + lane] = coming.descriptors[v]; }
		#pragma unroll
		for (unsigned v = 0; v < value_loads; v++) { reinterpret_cast<sixteen_bytes*>(staged_values)[v * 64 + lane] = coming.values[v]; }
		wave64::lds_fence();
	}
};

/// Writes tile outputs [0, tile_len) at `out`; values are first_run_data[rank(position)].
/// `staged` (may be null) is a wave-private LDS area of value_staging<ElementSize>::capacity elements;
/// `num_runs_in_tile` = marks + 1.
/// AlreadyStaged: the caller has put the tile's run values into `staged` itself (next to its own loads of the run
/// descriptors, instead of one more dependent round trip here)
/// The cache policy of the stores of a tile that is expanded from STAGED values (no load between its stores, or after them; the look-up
/// path loads between its stores and wants the quick acknowledgement of `nt`).  By measurement, every instantiation under both policies
/// (2^30 outputs, mean run 32; write-through against nt): 1-byte elements -3...-8 points (a wavefront's whole tile is 4 KiB: the
/// longer acknowledgement is most of its life), 2-byte +4...+7 (RPE with relative positions: -4), staged 4-byte +5...+7, 8-byte +4...+9
/// (32-bit values with 8-byte descriptors: -11...-13 while held to 4 workgroups per CU, +5...+7 without that cap).
template <unsigned ElementSize, unsigned RunDescriptorSize, bool RelativePositions = false>
constexpr int staged_store_policy()
{
	return (ElementSize == 1 || (ElementSize == 2 && RelativePositions))
		? (int) memory_access::non_temporal : (int) GIDDY_RUN_STAGED_STORE_POLICY;
}

/// a lane's group to an address aligned for an element only: one wide store all the same (gfx950 takes it; the lines it lies across are written in part)
template <typename T, unsigned N>
__device__ __forceinline__ void store_group_at_any_alignment(T* __restrict__ p, const memory_access::group<T, N>& g)
{
	constexpr unsigned bytes = N * sizeof(T);
	constexpr unsigned piece = bytes >= 16 ? 16 : bytes;
	typedef typename memory_access::chunk<piece>::type loose_piece_t __attribute__((aligned(sizeof(T))));
	using piece_t = typename memory_access::chunk<piece>::type;
	#pragma unroll
	for (unsigned i = 0; i < bytes / piece; i++) { reinterpret_cast<loose_piece_t*>(p)[i] = reinterpret_cast<const piece_t*>(&g)[i]; }
}

/// StagedStores: memory_access::store_policy of the 16-byte stores when the tile's values are staged
template <unsigned ElementSize, bool Staged, bool AlreadyStaged = false, int StagedStores = memory_access::non_temporal>
__device__ __forceinline__ void expand(
	uint_t<ElementSize>* __restrict__ out, uint32_t tile_len, const tile_entry* table,
	const uint_t<ElementSize>* __restrict__ first_run_data, unsigned lane,
	uint_t<ElementSize>* staged = nullptr, uint32_t num_runs_in_tile = 0)
{
	using namespace memory_access;
	using element_type = uint_t<ElementSize>;
	constexpr unsigned E = 16 / ElementSize;
	const bool aligned = is_aligned(out, 16);
	const bool use_staged = Staged && staged != nullptr && num_runs_in_tile <= value_staging<ElementSize>::capacity;
	if (use_staged && !AlreadyStaged) {
		// four loads per lane in flight at a time (unconditional: a lane past the last run re-reads that one), not one after the other
		for (uint32_t base = lane; base < num_runs_in_tile + lane; base += 4 * 64) {
			element_type fetched[4];
			#pragma unroll
			for (unsigned j = 0; j < 4; j++) { const uint32_t i = base + 64 * j; fetched[j] = first_run_data[i < num_runs_in_tile ? i : num_runs_in_tile - 1]; }
			#pragma unroll
			for (unsigned j = 0; j < 4; j++) { const uint32_t i = base + 64 * j; if (i < num_runs_in_tile) { staged[i] = fetched[j]; } }
		}
	}
	if (use_staged) { wave64::lds_fence(); }

	// Rows — one 16-byte store per lane, 1 KiB per wave instruction — begin on 512-byte boundaries of the OUTPUT, not of the tile: the
	// first row is short by `pad` elements (its first lanes idle).  A tile that begins 128 or 256 bytes past such a boundary cost 7-9
	// points with rows laid from the tile's start (every 512-byte block written by two instructions); 512 bytes on, nothing.
	// (4- and 8-byte elements: for 1- and 2-byte columns — 16 and 8 outputs per lane, tiles of 4 and 8 KiB — the bookkeeping costs 1-3 points even where
	// the pad is 0, and a displaced output was never their problem)
	const uint32_t pad = (ElementSize >= 4 && aligned) ? (uint32_t) ((reinterpret_cast<uintptr_t>(out) & 511u) / ElementSize) : 0u;
	for (uint32_t row_start = 0; row_start < tile_len + pad; row_start += 64 * E) {
		const uint32_t p = row_start + lane * E - pad;   // (wraps below zero for the first row's idle lanes)
		if (p >= tile_len) { continue; }
		const tile_entry entry = table[p >> 5];   // E divides 32: all E positions share the word
		auto rank_of = [&](unsigned j) {
			const unsigned bit = (p + j) & 31;
			return entry.rank_before + (uint32_t) __builtin_popcount(entry.bits & (0xFFFFFFFFu >> (31 - bit)));
		};
		group<element_type, E> values;
		bool done = false;
		if constexpr (Staged && ElementSize == 1) {
			// A lane's 16 one-byte outputs: when at most 8 runs cover them (any mean run length above ~2) they are a byte
			// permutation of the 8 staged values from the lane's first run on — three LDS dword reads and one
			// v_perm_b32 per output dword, instead of sixteen masked popcounts and sixteen byte-sized LDS reads.
			const unsigned first_bit = p & 31;   // 0 or 16
			const uint32_t starts = (entry.bits >> first_bit >> 1) & 0x7FFFu;   // bit i: a run starts at output p + 1 + i
			if (use_staged && wave64::ballot(__builtin_popcount(starts) > 7) == 0) {
				const uint32_t first_rank = entry.rank_before + (uint32_t) __builtin_popcount(entry.bits & (0xFFFFFFFFu >> (31 - first_bit)));
				const uint32_t* words = reinterpret_cast<const uint32_t*>(staged) + (first_rank >> 2);
				const uint32_t w0 = words[0], w1 = words[1], w2 = words[2];
				const uint32_t low = __builtin_amdgcn_alignbyte(w1, w0, first_rank & 3), high = __builtin_amdgcn_alignbyte(w2, w1, first_rank & 3);
				uint32_t runs_before = 0;   // runs starting at the lane's outputs 1 .. 4k - 1, replicated in all four bytes
				#pragma unroll
				for (unsigned k = 0; k < 4; k++) {
					// bit t: a run starts at output 4k + t (the lane's first run, at or before output 0, is not counted)
					const uint32_t nibble = k == 0 ? (starts << 1) & 0xEu : (starts >> (4 * k - 1)) & 0xFu;
					uint32_t selector = (nibble * 0x00204081u) & 0x01010101u;   // bit t -> byte t
					selector += selector << 8;
					selector += selector << 16;                                 // byte t: starts at outputs 4k .. 4k + t
					values.words[k] = __builtin_amdgcn_perm(high, low, selector + runs_before);
					runs_before += (uint32_t) __builtin_popcount(nibble) * 0x01010101u;
				}
				done = true;
			}
		}
		if constexpr (Staged && ElementSize == 2) {
			// ... and a lane's 8 two-byte outputs, when at most 4 runs cover them, of the 4 staged values from its first run on
			const unsigned first_bit = p & 31;   // 0, 8, 16 or 24
			const uint32_t starts = (entry.bits >> first_bit >> 1) & 0x7Fu;   // bit i: a run starts at output p + 1 + i
			if (use_staged && wave64::ballot(__builtin_popcount(starts) > 3) == 0) {
				const uint32_t first_rank = entry.rank_before + (uint32_t) __builtin_popcount(entry.bits & (0xFFFFFFFFu >> (31 - first_bit)));
				const uint32_t* words = reinterpret_cast<const uint32_t*>(staged) + (first_rank >> 1);
				const uint32_t w0 = words[0], w1 = words[1], w2 = words[2];
				const uint32_t low = __builtin_amdgcn_alignbyte(w1, w0, 2 * (first_rank & 1)), high = __builtin_amdgcn_alignbyte(w2, w1, 2 * (first_rank & 1));
				uint32_t selector = 0x01000100u;   // bytes {2r, 2r + 1} of value r, for both outputs of a dword; r = 0 so far
				#pragma unroll
				for (unsigned k = 0; k < 4; k++) {
					const uint32_t at_even = k == 0 ? 0 : (starts >> (2 * k - 1)) & 1u, at_odd = (starts >> (2 * k)) & 1u;   // a run starts at output 2k / 2k + 1
					selector += at_even * 0x02020202u;
					values.words[k] = __builtin_amdgcn_perm(high, low, selector + at_odd * 0x02020000u);
					selector += at_odd * 0x02020202u;
				}
				done = true;
			}
		}
		if (done) { }
		else if (use_staged) {
			// one output dword (4 / ElementSize elements) at a time, so that at most that many ranks and looked-up
			// values are alive at once (all 16 of a 1-byte column at once cost two wavefronts per SIMD of occupancy)
			// (wide elements: a lane's 2 or 4 look-ups all at once instead of one LDS round trip after the other)
			constexpr unsigned per_dword = ElementSize >= 4 ? E : 4 / (ElementSize > 4 ? 4 : ElementSize);
			#pragma unroll
			for (unsigned first = 0; first < E; first += per_dword) {
				#pragma unroll
				for (unsigned j = first; j < first + per_dword && j < E; j++) { values[j] = staged[rank_of(j)]; }
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		else {
			uint32_t ranks[E];
			#pragma unroll
			for (unsigned j = 0; j < E; j++) { ranks[j] = rank_of(j); }
			element_type first = first_run_data[ranks[0]];
			if (ranks[E - 1] == ranks[0]) {
				#pragma unroll
				for (unsigned j = 0; j < E; j++) { values[j] = first; }
			}
			else {
				values[0] = first;
				#pragma unroll
				for (unsigned j = 1; j < E; j++) { values[j] = first_run_data[ranks[j]]; }
			}
		}
		if (aligned && p + E <= tile_len) {
			// (a tile expanded from staged values issues no load between its stores, or after them: write-through stores)
			if (use_staged) { store_group<StagedStores>(out + p, values); }
			else { store_group(out + p, values); }
		}
		else if (p + E <= tile_len) { store_group_at_any_alignment(out + p, values); }   // (an output aligned for its elements only: 34 -> see DESIGN.md section 4, "what is still a cliff")
		else {
			#pragma unroll
			for (unsigned j = 0; j < E; j++) { if (p + j < tile_len) { out[p + j] = values[j]; } }
		}
	}
	wave64::lds_fence();   // the table (and the staged values) may be rewritten for the next tile after this
}

} // namespace run_expansion
} // namespace decompression
} // namespace kernels
} // namespace cuda
