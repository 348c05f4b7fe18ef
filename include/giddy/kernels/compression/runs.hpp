// Device-side compressors for the two run encodings (formats: libgiddy_amd/formats.py::run_length_encoding,
// ::run_position_encoding; reference: run_length_encoding.cuh:271-326, run_position_encoding.cuh:255-308).
//
// A run starts wherever an element differs from its predecessor.  Run-position encoding stores run k's value, its start position
// (absolute, or relative to its segment) and, for every segment, the index of the run covering its first element.  Run-LENGTH
// encoding stores lengths, and a length field holds at most `max_run_length`: a longer run is cut into pieces of that length
// (formats._runs_of), so the index of a piece depends on the lengths of the runs before it, not only on their number.
// Both in ONE launch each since round 3 (the column is read once): see `one_pass` below and compression/chained.hpp.
#pragma once
#include "giddy/kernels/compression/chained.hpp"

namespace cuda {
namespace kernels {
namespace compression {
namespace runs {

// ---- one pass (compression/chained.hpp) ---------------------------------------------------------------------------------------
//
// A workgroup's tile is 16 384 elements, 4096 per wavefront; a wavefront reads its 4096 as ROWS of 16 bytes per lane (coalesced) and a
// lane keeps ONE bit per element it saw — 64 bits: "a run starts here" — not the elements: a tile lives some 25 microseconds, most of it
// waiting (for its loads, for the tiles before it, for the values it reads again), so what is in flight must be many tiles — 130
// registers per lane leave room for 4 workgroups per CU and SIMD — and the few elements that become run values are read again
// afterwards, out of the L2 / the Infinity Cache that they passed through microseconds before.
// Output order is element order = row by row, lane by lane: a run's index is its tile's base + the runs of the wavefronts before +
// of the rows before + of the lanes before in its row (one packed DPP scan for four rows) + of the lane's own earlier bits in the row.

namespace one_pass {

enum : unsigned { per_lane = 64, wave_elements = 64 * per_lane, tile = waves_per_block * wave_elements };
using mask_t = std::conditional<(per_lane > 32), uint64_t, uint32_t>::type;   // a lane's bits, one per element it saw
enum : unsigned {
	staged_runs = 2048,          // runs (pieces) of a tile whose values and descriptors wait in LDS for the tile's base
	tiles_per_link = 4,          // consecutive tiles a workgroup of encode_run_positions takes as ONE link of the chain (round 5, 2^28 elements: 2 / 4 / 8 / 16 tiles: 0.46 / 0.45 / 0.46-0.50 / 0.53 ms)
	values_read_together = 4,    // run values a lane reads again at a time: one round trip
	waves_per_simd = 4,          // (6 and 8 — the compiler then spills — measured slower: profiles/r03_compress_bench.txt)
};
__device__ __forceinline__ unsigned count_bits(mask_t m) { return (unsigned) __builtin_popcountll((unsigned long long) m); }
__device__ __forceinline__ unsigned lowest_bit(mask_t m) { return (unsigned) __builtin_ctzll((unsigned long long) m); }
__device__ __forceinline__ unsigned highest_bit(mask_t m) { return 63u - (unsigned) __builtin_clzll((unsigned long long) m); }

template <unsigned ElementSize>
struct shape {
	static constexpr unsigned E = 16 / ElementSize;        // elements per lane per row (2 ... 16)
	static constexpr unsigned rows = per_lane / E;         // 16 ... 2
	static constexpr unsigned row_elements = 64 * E;
	static constexpr mask_t row_mask = ((mask_t) 1 << E) - 1;
	static constexpr unsigned groups = (rows + 3) / 4;     // uint64 words of four 16-bit per-row fields
};

template <typename T>
__device__ __forceinline__ T from_the_lane_before(T value, T into_first_lane)
{
	using carrier = typename std::conditional<(sizeof(T) > 4), uint64_t, uint32_t>::type;
	return (T) wave64::shift_up_by_one((carrier) value, (carrier) into_first_lane);
}

/// Bit r * E + e of the result: a run starts at element wave_base + r * row_elements + lane * E + e.  `last_value`: the wavefront's
/// last element (meaningful when all its elements lie inside the column).  The rows are read 32 elements per lane at a time (what is
/// in flight per wavefront: 8 KiB) — `Whole`: all of them inside the column and 16-byte aligned: loads without a branch between them.
template <typename T, bool Whole>
__device__ __forceinline__ mask_t run_starts_of_wavefront(const T* __restrict__ input, uint64_t wave_base, uint64_t n, unsigned lane, T& last_value)
{
	using S = shape<sizeof(T)>;
	constexpr unsigned batch_rows = (sizeof(T) <= 4 ? per_lane : 32) / S::E, batches = S::rows / batch_rows;   // (64 registers of data at most)
	T carry = 0;   // what the first lane's first element of a row is compared with
	mask_t starts = 0;
	#pragma unroll
	for (unsigned batch = 0; batch < batches; batch++) {
		T own[batch_rows][S::E];
		#pragma unroll
		for (unsigned i = 0; i < batch_rows; i++) {
			const uint64_t pos = wave_base + (batch * batch_rows + i) * S::row_elements + lane * S::E;
			if (Whole) {
				const memory_access::group<T, S::E> loaded = memory_access::load_group<T, S::E>(input + pos);
				#pragma unroll
				for (unsigned e = 0; e < S::E; e++) { own[i][e] = loaded[e]; }
			}
			else {
				#pragma unroll
				for (unsigned e = 0; e < S::E; e++) { own[i][e] = pos + e < n ? input[pos + e] : (T) 0; }
			}
		}
		if (batch == 0) { carry = wave_base == 0 ? (T) ~own[0][0] : input[wave_base - 1]; }
		#pragma unroll
		for (unsigned i = 0; i < batch_rows; i++) {
			const unsigned r = batch * batch_rows + i;
			const uint64_t pos = wave_base + r * S::row_elements + lane * S::E;
			const T left = from_the_lane_before(own[i][S::E - 1], carry);
			#pragma unroll
			for (unsigned e = 0; e < S::E; e++) {
				starts |= (mask_t) ((Whole || pos + e < n) && own[i][e] != (e == 0 ? left : own[i][e - 1])) << (r * S::E + e);
			}
			carry = wave64::read_last_lane(own[i][S::E - 1]);
		}
		__builtin_amdgcn_sched_barrier(0);   // (the next batch's loads are not to be hoisted above this one's use: registers)
	}
	last_value = carry;
	return starts;
}

template <typename T>
__device__ __forceinline__ mask_t run_starts_of_wavefront(const T* __restrict__ input, uint64_t wave_base, uint64_t n, unsigned lane, T& last_value)
{
	if (memory_access::is_aligned(input, 16) && wave_base + wave_elements <= n) { return run_starts_of_wavefront<T, true>(input, wave_base, n, lane, last_value); }
	return run_starts_of_wavefront<T, false>(input, wave_base, n, lane, last_value);
}

/// Where the lane's bits go in the output: `offsets` = per row, (set bits of the wavefront's earlier rows) + (of the earlier lanes in
/// this row), as 16-bit fields, four rows to a word; returns the wavefront's number of set bits.
template <unsigned ElementSize>
__device__ __forceinline__ uint32_t rank_rows(mask_t bits, uint64_t (&offsets)[shape<ElementSize>::groups])
{
	using S = shape<ElementSize>;
	uint32_t running = 0;
	#pragma unroll
	for (unsigned g = 0; g < S::groups; g++) {
		uint64_t counts = 0;
		#pragma unroll
		for (unsigned r = 4 * g; r < 4 * g + 4 && r < S::rows; r++) { counts |= (uint64_t) count_bits((bits >> (r * S::E)) & S::row_mask) << (16 * (r % 4)); }
		const uint64_t inclusive = wave64::inclusive_sum(counts), totals = wave64::read_last_lane(inclusive), exclusive = inclusive - counts;
		uint64_t fields = 0;
		#pragma unroll
		for (unsigned r = 4 * g; r < 4 * g + 4 && r < S::rows; r++) {
			fields |= (uint64_t) (running + (uint32_t) ((exclusive >> (16 * (r % 4))) & 0xFFFF)) << (16 * (r % 4));
			running += (uint32_t) ((totals >> (16 * (r % 4))) & 0xFFFF);
		}
		offsets[g] = fields;
	}
	return running;
}

template <unsigned Groups>
__device__ __forceinline__ uint32_t offset_of_row(const uint64_t (&offsets)[Groups], unsigned r)
{
	uint64_t word = offsets[0];
	#pragma unroll
	for (unsigned g = 1; g < Groups; g++) { word = r / 4 == g ? offsets[g] : word; }
	return (uint32_t) ((word >> (16 * (r % 4))) & 0xFFFF);
}

/// an element's place in its segment, given its wavefront's first element's (`wave_into_segment` < L) and its offset from that (< 2048)
__device__ __forceinline__ uint64_t into_segment(uint64_t wave_into_segment, uint32_t offset, uint64_t L)
{
	const uint64_t x = wave_into_segment + offset;
	return x < L ? x : (L >= wave_elements ? x - L : x % L);
}

/// ... and how many segment starts lie between the wavefront's first element (exclusive) and it (inclusive)
__device__ __forceinline__ uint64_t segments_past(uint64_t wave_into_segment, uint32_t offset, uint64_t L)
{
	const uint64_t x = wave_into_segment + offset;
	return x < L ? 0 : (L >= wave_elements ? 1 : x / L);
}

/// Run-position encoding in ONE launch: run k (in order of position) -> run_data[k], run_start_positions[k]; position_anchors[s] =
/// the run covering s * L; *num_element_runs = the number of runs.  `states`: one zeroed uint64 per tile.
template <unsigned IndexSize, unsigned UncompressedSize, unsigned PositionSize, bool PositionsAreRelative, unsigned RunIndexSize>
__global__ void __launch_bounds__(threads_per_block, waves_per_simd) encode_run_positions(
	uint_t<UncompressedSize>*        __restrict__  run_data,
	uint_t<PositionSize>*            __restrict__  run_start_positions,
	uint_t<RunIndexSize>*            __restrict__  position_anchors,
	uint64_t*                        __restrict__  num_element_runs,
	const uint_t<UncompressedSize>*  __restrict__  input,
	uint64_t*                        __restrict__  states,
	size_type_by_index_size<IndexSize>             segment_length,
	size_type_by_index_size<IndexSize>             length,
	uint64_t                                       capacity,   // runs the per-run arrays have room for: runs beyond it are counted, not stored
	uint32_t                                       patience,   // polls a tile waits for the tiles before it (chained.hpp, "Progress")
	uint64_t                                       fault_tile) // chained::no_fault_tile; a test's tile that plays dead
{
	using T = uint_t<UncompressedSize>;
	using S = shape<UncompressedSize>;
	__shared__ uint32_t wave_totals[waves_per_block];
	__shared__ uint64_t slot;
	__shared__ T staged_values[staged_runs];
	__shared__ uint_t<PositionSize> staged_positions[staged_runs];
	const uint64_t n = length, L = segment_length, num_tiles = (n + tile - 1) / tile;
	const unsigned lane = wave64::lane_id(), wave = wave64::index_in_block();
	// A workgroup is ONE link of the chain and takes `tiles_per_link` CONSECUTIVE tiles (round 4): it finds the run starts of all of them first —
	// a lane keeps one bit per element, eight registers for four tiles —, publishes the link's number of runs, and looks back ONCE while its
	// first tile's runs are being staged: a quarter of the links, each four times as long (16 384 links of 64 KiB: 433-439 us at 2^28 elements).
	constexpr unsigned K = tiles_per_link;
	const uint64_t num_links = (num_tiles + K - 1) / K;
	__shared__ uint32_t link_wave_totals[K][waves_per_block];
	for (uint64_t link = blockIdx.x; link < num_links; link += gridDim.x) {
	if (link == fault_tile) { return; }   // (a test: this link is never heard of)
	mask_t starts_of_tile[K];
	#pragma unroll
	for (unsigned k = 0; k < K; k++) {
		const uint64_t wave_base = (link * K + k) * tile + (uint64_t) wave * wave_elements;
		T last_value;
		starts_of_tile[k] = wave_base < n ? run_starts_of_wavefront<T>(input, wave_base, n, lane, last_value) : 0;
		const uint32_t of_wave = wave64::read_last_lane(wave64::inclusive_sum((uint32_t) count_bits(starts_of_tile[k])));
		if (lane == 0) { link_wave_totals[k][wave] = of_wave; }
	}
	__syncthreads();
	uint64_t link_total = 0;
	#pragma unroll
	for (unsigned k = 0; k < K; k++) { for (unsigned w = 0; w < waves_per_block; w++) { link_total += link_wave_totals[k][w]; } }
	chained::announce(states, link, link_total);   // (the links after this one wait for it: said as soon as it is known)
	uint64_t runs_before_tile = 0;   // known from the first tile's look-back on
	#pragma unroll
	for (unsigned k = 0; k < K; k++) {
		const uint64_t t = link * K + k;
		if (t >= num_tiles) { break; }
		const uint64_t wave_base = t * tile + (uint64_t) wave * wave_elements;
#ifdef GIDDY_CHAINED_STAMPS   // (measurements: when a tile started, had counted, knew its base, was done — 100 MHz ticks, behind the states)
		uint64_t* const stamps = states + num_tiles + 4 * t;
		if (threadIdx.x == 0) { stamps[0] = __builtin_amdgcn_s_memrealtime(); }
#endif
		const mask_t starts = starts_of_tile[k];
		uint64_t offsets[S::groups];
		(void) rank_rows<UncompressedSize>(starts, offsets);
		uint32_t wave_before = 0, tile_total = 0;
		#pragma unroll
		for (unsigned w = 0; w < waves_per_block; w++) { const uint32_t of_wave = link_wave_totals[k][w]; wave_before += w < wave ? of_wave : 0; tile_total += of_wave; }
		// A tile of at most `staged_runs` runs: their values (read again, out of the L2 they just passed through) and positions go to LDS at their
		// rank IN THE TILE now — while the tiles before this one are still being waited for — and leave as coalesced copies once the tile knows its
		// base.  (With the re-reads after the chain a tile spent 10 of its 27 microseconds there.)
		const bool staged = tile_total <= staged_runs;   // (block-uniform)
		const uint64_t wave_into_segment = wave_base < n ? wave_base % L : 0, wave_segment = wave_base / L;
		auto for_each_run_of_the_lane = [&](auto&& emit) {   // emit(rank in the tile, offset in the wavefront, value)
			constexpr unsigned together = values_read_together;   // (values read that many at a time: one round trip)
			for (mask_t rest = starts; rest != 0; ) {
				uint32_t offset[together]; T value[together]; bool have[together];
				mask_t walk = rest;
				#pragma unroll
				for (unsigned i = 0; i < together; i++) {
					have[i] = walk != 0;
					const unsigned bit = have[i] ? lowest_bit(walk) : 0;
					offset[i] = (bit / S::E) * S::row_elements + lane * S::E + bit % S::E;
					value[i] = have[i] ? input[wave_base + offset[i]] : (T) 0;
					walk &= walk - 1;
				}
				#pragma unroll
				for (unsigned i = 0; i < together; i++) {
					if (have[i]) {
						const unsigned bit = lowest_bit(rest), r = bit / S::E;
						emit(wave_before + offset_of_row(offsets, r) + count_bits(starts & (((mask_t) 1 << bit) - 1) & (S::row_mask << (r * S::E))), offset[i], value[i]);
						rest &= rest - 1;
					}
				}
			}
		};
		auto position_of = [&](uint32_t offset) { return (uint_t<PositionSize>) (PositionsAreRelative ? into_segment(wave_into_segment, offset, L) : wave_base + offset); };
		if (k != 0) { __syncthreads(); }   // (the staging area has been copied out for the tile before)
		if (staged) {
			for_each_run_of_the_lane([&](uint32_t rank, uint32_t offset, T value) { staged_values[rank] = value; staged_positions[rank] = position_of(offset); });
		}
#ifdef GIDDY_CHAINED_STAMPS
		if (threadIdx.x == 0) { stamps[1] = __builtin_amdgcn_s_memrealtime(); }
#endif
		if (k == 0) {   // the link's first tile: its runs are staged — now, and once per link, the look-back (barriers inside: the staging is complete)
			runs_before_tile = chained::exclusive_base(states, link, num_links, link_total, &slot, num_element_runs, true, patience);
			if (runs_before_tile == chained::failed_base) { return; }   // (the chain broke before this link: every thread sees it; the last link has said so in *num_element_runs)
		}
		else { __syncthreads(); }   // (the staging is complete)
		const uint64_t tile_base = runs_before_tile;
		runs_before_tile += tile_total;
		const uint64_t base = tile_base + wave_before;   // ... before this wavefront's elements
#ifdef GIDDY_CHAINED_STAMPS
		if (threadIdx.x == 0) { stamps[2] = __builtin_amdgcn_s_memrealtime(); }
#endif
		if (staged) {
			for (uint32_t i = threadIdx.x; i < tile_total; i += threads_per_block) {
				if (tile_base + i < capacity) { run_data[tile_base + i] = staged_values[i]; run_start_positions[tile_base + i] = staged_positions[i]; }
			}
		}
		if (wave_base >= n) { continue; }
		if (!staged) {
			for_each_run_of_the_lane([&](uint32_t rank, uint32_t offset, T value) {
				const uint64_t k = tile_base + rank;
				if (k < capacity) { run_data[k] = value; run_start_positions[k] = position_of(offset); }
			});
		}
		// the anchors of the segments that begin among this lane's elements: the run covering a segment's first element = the last that starts at or before it
		uint64_t row_into_segment = wave_into_segment;   // (wave-uniform) of the row's first element: a row without a segment's first element is skipped whole
		#pragma nounroll
		for (unsigned r = 0; r < S::rows; r++) {
			const bool row_has_a_boundary = row_into_segment == 0 || row_into_segment + S::row_elements > L;
			row_into_segment += S::row_elements;
			if (row_into_segment >= L) { row_into_segment = L >= S::row_elements ? row_into_segment - L : row_into_segment % L; }
			if (!row_has_a_boundary) { continue; }
			const uint32_t offset = r * S::row_elements + lane * S::E;
			const uint64_t here = into_segment(wave_into_segment, offset, L);
			for (uint64_t e = here == 0 ? 0 : L - here; e < S::E && wave_base + offset + e < n; e += L) {
				const unsigned bit = r * S::E + (unsigned) e;
				const uint64_t run = base + offset_of_row(offsets, r) + count_bits(starts & ((((mask_t) 1 << bit) << 1) - 1) & (S::row_mask << (r * S::E))) - 1;
				position_anchors[wave_segment + segments_past(wave_into_segment, offset + (uint32_t) e, L)] = (uint_t<RunIndexSize>) run;
			}
		}
#ifdef GIDDY_CHAINED_STAMPS
		if (threadIdx.x == 0) { stamps[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
	}
	__syncthreads();   // (the link's tables are rewritten by the workgroup's next link)
	}
}

/// Run-LENGTH encoding in ONE launch.  A length field holds at most M = `max_run_length`, so a natural run is stored as pieces of M
/// and a remainder (formats._runs_of).  A piece is emitted where it ENDS — after element e when e + 1 starts a run, ends the
/// column, or e is the M-th element of its piece — by the lane that saw e: it needs to know where e's run started (possibly
/// tiles ago) but not where it ends (possibly tiles ahead).  Two chains: `start_states` hands the start of the run that reaches
/// into a tile from the tiles before it (a tile with a run start of its own publishes it at once, the others what they
/// inherited), `count_states` the number of pieces that end before the tile.  Both zeroed, one uint64 per tile each.
/// Segment s's anchor: the piece holding element s * L and the element's offset into it.
template <unsigned IndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize>
__global__ void __launch_bounds__(threads_per_block, waves_per_simd) encode_run_lengths(
	uint_t<UncompressedSize>*        __restrict__  run_data,
	uint_t<RunLengthSize>*           __restrict__  run_lengths,
	uint_t<RunIndexSize>*            __restrict__  position_anchors,
	uint_t<RunLengthSize>*           __restrict__  intra_run_anchor_offsets,
	uint64_t*                        __restrict__  num_element_runs,
	const uint_t<UncompressedSize>*  __restrict__  input,
	uint64_t*                        __restrict__  start_states,
	uint64_t*                        __restrict__  count_states,
	size_type_by_index_size<IndexSize>             segment_length,
	size_type_by_index_size<IndexSize>             length,
	uint64_t                                       max_run_length,
	uint64_t                                       capacity,   // pieces run_data / run_lengths have room for
	uint32_t                                       patience,   // polls a tile waits for the tiles before it (chained.hpp, "Progress")
	uint64_t                                       fault_tile) // chained::no_fault_tile; a test's tile that plays dead
{
	using T = uint_t<UncompressedSize>;
	using S = shape<UncompressedSize>;
	__shared__ uint32_t wave_totals[waves_per_block];
	__shared__ uint32_t last_starts[waves_per_block];
	__shared__ uint64_t slot, inherited;
	__shared__ T staged_values[staged_runs];
	__shared__ uint_t<RunLengthSize> staged_lengths[staged_runs];
	const uint64_t n = length, L = segment_length, M = max_run_length, num_tiles = (n + tile - 1) / tile;
	const unsigned lane = wave64::lane_id(), wave = wave64::index_in_block();
	for (uint64_t t = blockIdx.x; t < num_tiles; t += gridDim.x) {
		if (t == fault_tile) { return; }   // (a test: this tile is never heard of)
		const uint64_t tile_begin = t * tile, wave_base = tile_begin + (uint64_t) wave * wave_elements;
#ifdef GIDDY_CHAINED_STAMPS   // (measurements: started / scanned for starts / knows what reaches in / counted / knows its base / done — behind the states)
		uint64_t* const stamps = count_states + num_tiles + 6 * t;
		if (threadIdx.x == 0) { stamps[0] = __builtin_amdgcn_s_memrealtime(); }
#endif
		T last_value = 0;
		const mask_t starts = wave_base < n ? run_starts_of_wavefront<T>(input, wave_base, n, lane, last_value) : 0;
		// does a run start right after the wavefront's last element?
		const bool run_starts_after_wave = wave_base + wave_elements < n && input[wave_base + wave_elements] != last_value;

		// the last start before each of the lane's rows, as 1 + its offset in the TILE (0: none in the tile): row-major maximum; 16-bit
		// fields, four rows to a word
		uint64_t before_rows[S::groups] = { };
		uint32_t carry = 0;   // (wave-uniform: the last start in the rows before)
		#pragma unroll
		for (unsigned r = 0; r < S::rows; r++) {
			const uint32_t mine = (uint32_t) ((starts >> (r * S::E)) & S::row_mask);
			const uint32_t my_last = mine == 0 ? 0 : wave * wave_elements + r * S::row_elements + lane * S::E + (31 - __builtin_clz(mine)) + 1;
			const uint32_t through_me = wave64::inclusive_max(my_last);
			const uint32_t lanes_before = wave64::shift_up_by_one(through_me, 0u);
			before_rows[r / 4] |= (uint64_t) (lanes_before > carry ? lanes_before : carry) << (16 * (r % 4));
			const uint32_t of_row = wave64::read_last_lane(through_me);
			carry = of_row > carry ? of_row : carry;
		}
#ifdef GIDDY_CHAINED_STAMPS
		if (threadIdx.x == 0) { stamps[1] = __builtin_amdgcn_s_memrealtime(); }
#endif
		__syncthreads();   // (the tables may still be read for the tile before)
		if (lane == 0) { last_starts[wave] = carry; }
		__syncthreads();
		uint32_t waves_before = 0, in_tile = 0;
		#pragma unroll
		for (unsigned w = 0; w < waves_per_block; w++) {
			const uint32_t of_wave = last_starts[w];
			if (w < wave && of_wave > waves_before) { waves_before = of_wave; }
			if (of_wave > in_tile) { in_tile = of_wave; }
		}
		if (wave == 0) {
			// the first chain's states: 0 nothing yet; flag `aggregate`: no start in the tile, what reaches into it not known yet;
			// flag `inclusive`: the last start at or before the tile's end
			if (in_tile != 0 && lane == 0) { chained::publish(start_states, t, tile_begin + in_tile - 1, chained::inclusive); }
			uint64_t reaching_in = 0;
			const bool needed = t != 0 && (wave64::read_first_lane((uint32_t) starts) & 1) == 0;   // the tile's first element continues a run
			if (needed) {
				if (in_tile == 0 && lane == 0) { chained::publish(start_states, t, 0, chained::aggregate); }
				uint64_t end = t;
				bool found = false;
				uint32_t polls = 0;
				bool broken = !chained::wait_for(start_states, t - 1, polls, patience);   // (one word first: chained.hpp)
				while (!found && !broken) {
					const uint64_t word = lane < end ? chained::peek(start_states, end - 1 - lane) : (uint64_t) chained::inclusive;   // (before the column: start 0)
					const uint64_t flag = word & chained::flag_mask;
					const uint64_t known = wave64::ballot(flag == chained::inclusive), silent = wave64::ballot(flag == chained::nothing || flag == chained::failed);
					const unsigned first_known = known != 0 ? (unsigned) __builtin_ctzll(known) : 64u, first_silent = silent != 0 ? (unsigned) __builtin_ctzll(silent) : 64u;
					if (first_known < first_silent) { reaching_in = wave64::read_lane(word, first_known) >> 2; found = true; }
					else if (first_silent == 64u) { end -= 64; }   // 64 tiles without a start, none of which knows yet: further back
					else if ((wave64::read_lane(word, first_silent) & chained::flag_mask) == chained::failed || ++polls > patience) { broken = true; }   // a tile before this one gave up, or is not heard of
					else { __builtin_amdgcn_s_sleep(2); }
				}
				if (broken) { reaching_in = chained::failed_base; if (lane == 0) { chained::publish(start_states, t, 0, chained::failed); } }
				else if (in_tile == 0 && lane == 0) { chained::publish(start_states, t, reaching_in, chained::inclusive); }
			}
			if (lane == 0) { inherited = reaching_in; }
		}
		__syncthreads();
		const uint64_t start_before_tile = inherited;
		if (start_before_tile == chained::failed_base) {   // the chain of run starts broke: this tile gives up in the chain of counts as well (every thread takes this way)
			if (threadIdx.x == 0) { chained::give_up(count_states, t, num_tiles, num_element_runs); }
			return;
		}
#ifdef GIDDY_CHAINED_STAMPS
		if (threadIdx.x == 0) { stamps[2] = __builtin_amdgcn_s_memrealtime(); }
#endif

		// `ends`: bit r * E + e: a piece ends with that element.  offset_at_row(r): the distance of the lane's first element of row r from its
		// piece's first (where no run starts there), modulo M — from where the run that covers it started
		using offset_type = typename std::conditional<(RunLengthSize > 4), uint64_t, uint32_t>::type;
		const offset_type last_offset = (offset_type) (M - 1);
		auto offset_at_row = [&](unsigned r) -> offset_type {
			const uint32_t in_wave = offset_of_row(before_rows, r), last_start_before = in_wave > waves_before ? in_wave : waves_before;   // in the tile, 1-based; 0: none
			const uint64_t run_start = last_start_before != 0 ? tile_begin + last_start_before - 1 : start_before_tile;
			const uint64_t pos = wave_base + r * S::row_elements + lane * S::E;
			const uint64_t distance = pos < n ? pos - run_start : 0;
			return (offset_type) (distance < M ? distance : distance % M);
		};
		// (wave-uniform: can a piece fill up among this wavefront's elements at all?  Not if even the run that reaches into it — begun at or after
		// the last start known before the wavefront — stays shorter than M to the wavefront's end: then no row needs its offset here)
		const uint64_t start_before_wave = waves_before != 0 ? tile_begin + waves_before - 1 : start_before_tile;
		const bool a_piece_may_fill_up = wave_base + wave_elements - start_before_wave >= M;
		mask_t ends = 0;
		#pragma nounroll
		for (unsigned r = 0; r < S::rows; r++) {
			const uint64_t pos = wave_base + r * S::row_elements + lane * S::E;
			const uint32_t row_bits = (uint32_t) ((starts >> (r * S::E)) & S::row_mask);
			// whether the element after the lane's row piece starts a run: the next lane's first bit of the row; for the last lane the next row's first lane's
			const uint32_t first_bit_of_next_lane = __shfl_down(row_bits & 1u, 1, 64);
			const uint32_t first_bit_of_next_row = r + 1 < S::rows ? wave64::read_first_lane((uint32_t) ((starts >> ((r + 1 < S::rows ? r + 1 : r) * S::E)) & 1u)) : (uint32_t) run_starts_after_wave;
			const bool next_starts_after_row = (lane == 63 ? first_bit_of_next_row : first_bit_of_next_lane) != 0;
			offset_type offset = a_piece_may_fill_up ? offset_at_row(r) : 0;
			uint32_t row_ends = 0;
			if (pos + S::E < n && (uint64_t) offset + S::E <= (uint64_t) last_offset) {
				// no piece can fill up among these elements and the column goes on after them: a piece ends where the next element starts a run
				row_ends = (row_bits >> 1) | ((uint32_t) next_starts_after_row << (S::E - 1));
			}
			else {
				#pragma unroll
				for (unsigned e = 0; e < S::E; e++) {
					if ((row_bits >> e) & 1u) { offset = 0; }
					const bool next_starts = e + 1 < S::E ? ((row_bits >> (e + 1)) & 1u) != 0 : next_starts_after_row;
					const bool end_here = pos + e < n && (pos + e + 1 == n || next_starts || offset == last_offset);
					row_ends |= (uint32_t) end_here << e;
					offset = end_here ? 0 : offset + 1;
				}
			}
			ends |= (mask_t) row_ends << (r * S::E);
		}
		uint64_t offsets[S::groups];
		const uint32_t wave_total = rank_rows<UncompressedSize>(ends, offsets);
		if (lane == 0) { wave_totals[wave] = wave_total; }
		__syncthreads();
		uint32_t wave_before = 0, tile_total = 0;
		#pragma unroll
		for (unsigned w = 0; w < waves_per_block; w++) { const uint32_t of_wave = wave_totals[w]; wave_before += w < wave ? of_wave : 0; tile_total += of_wave; }
		// a piece begins where a run starts or right after a piece ended; within a lane's row piece, an element's offset into its piece is its distance
		// from the nearest such place at or before it — or, if there is none in the row piece, the row's own offset + its distance from the row's first
		// The pieces go out through LDS in WINDOWS of `staged_runs` ranks: the lane's pieces whose rank in the tile falls into the window are put there —
		// value (read again, out of the L2 it just passed through) and length — and the window leaves as coalesced copies.  The FIRST window is
		// filled before the look-back (while the tiles before this one are waited for: the write phase was 11 of a tile's 36 microseconds), the others —
		// tiles of more than `staged_runs` pieces: mean runs below 8 — after it.  One copy of the loop in the code: the window is its only variable.
		auto stage_window = [&](uint32_t window_first) {
			constexpr unsigned together = values_read_together;   // (values read that many at a time: one round trip)
			for (mask_t rest = wave_base < n ? ends : 0; rest != 0; ) {
				T value[together]; uint32_t rank[together]; bool have[together];
				mask_t walk = rest;
				#pragma unroll
				for (unsigned i = 0; i < together; i++) {
					const unsigned bit = walk != 0 ? lowest_bit(walk) : 0, r = bit / S::E;
					rank[i] = wave_before + offset_of_row(offsets, r) + count_bits(ends & (((mask_t) 1 << bit) - 1) & (S::row_mask << (r * S::E))) - window_first;   // (wraps below the window)
					have[i] = walk != 0 && rank[i] < staged_runs;
					value[i] = have[i] ? input[wave_base + r * S::row_elements + lane * S::E + bit % S::E] : (T) 0;
					walk &= walk - 1;
				}
				#pragma unroll
				for (unsigned i = 0; i < together; i++) {
					if (rest != 0) {
						const unsigned bit = lowest_bit(rest), r = bit / S::E, e = bit % S::E;
						if (have[i]) {
							const uint32_t begins = (uint32_t) ((((starts >> (r * S::E)) & S::row_mask) | (((ends >> (r * S::E)) & S::row_mask) << 1)) & ((2u << e) - 1) & S::row_mask);   // piece beginnings among the row's elements 0 .. e
							const offset_type piece_length = begins != 0 ? (offset_type) (e - (31 - __builtin_clz(begins)) + 1) : (offset_type) (offset_at_row(r) + e + 1);
							staged_values[rank[i]] = value[i];
							staged_lengths[rank[i]] = (uint_t<RunLengthSize>) piece_length;
						}
						rest &= rest - 1;
					}
				}
			}
		};
#ifdef GIDDY_CHAINED_STAMPS
		if (threadIdx.x == 0) { stamps[3] = __builtin_amdgcn_s_memrealtime(); }
#endif
		chained::announce(count_states, t, tile_total);   // (the tiles after this one wait for it: said as soon as it is known)
		uint64_t tile_base = 0;   // pieces that end before the tile: known after the first window has been staged
		for (uint32_t window_first = 0; window_first == 0 || window_first < tile_total; window_first += staged_runs) {
			if (window_first != 0) { __syncthreads(); }   // (the window before has been copied out)
			stage_window(window_first);
			if (window_first == 0) {
				tile_base = chained::exclusive_base(count_states, t, num_tiles, tile_total, &slot, num_element_runs, true, patience);   // (barriers inside: the window is complete)
				if (tile_base == chained::failed_base) { return; }   // (the chain broke before this tile: every thread sees it)
#ifdef GIDDY_CHAINED_STAMPS
				if (threadIdx.x == 0) { stamps[4] = __builtin_amdgcn_s_memrealtime(); }
#endif
			}
			else { __syncthreads(); }
			const uint32_t in_window = tile_total - window_first < staged_runs ? tile_total - window_first : staged_runs;
			for (uint32_t i = threadIdx.x; i < in_window; i += threads_per_block) {
				const uint64_t k = tile_base + window_first + i;
				if (k < capacity) { run_data[k] = staged_values[i]; run_lengths[k] = staged_lengths[i]; }
			}
		}
		const uint64_t base = tile_base + wave_before;   // ... before this wavefront's elements
		if (wave_base >= n) { continue; }
		const uint64_t wave_into_segment = wave_base % L, wave_segment = wave_base / L;
		uint64_t row_into_segment = wave_into_segment;   // (wave-uniform) of the row's first element: a row without a segment's first element is skipped whole
		#pragma nounroll
		for (unsigned r = 0; r < S::rows; r++) {
			const bool row_has_a_boundary = row_into_segment == 0 || row_into_segment + S::row_elements > L;
			row_into_segment += S::row_elements;
			if (row_into_segment >= L) { row_into_segment = L >= S::row_elements ? row_into_segment - L : row_into_segment % L; }
			if (!row_has_a_boundary) { continue; }
			const uint32_t offset_in_wave = r * S::row_elements + lane * S::E;
			const uint64_t here = into_segment(wave_into_segment, offset_in_wave, L);
			const offset_type at_row = offset_at_row(r);
			for (uint64_t e = here == 0 ? 0 : L - here; e < S::E && wave_base + offset_in_wave + e < n; e += L) {
				const unsigned bit = r * S::E + (unsigned) e;
				const uint64_t piece = base + offset_of_row(offsets, r) + count_bits(ends & (((mask_t) 1 << bit) - 1) & (S::row_mask << (r * S::E)));   // pieces that ended before the element
				const uint32_t begins = (uint32_t) ((((starts >> (r * S::E)) & S::row_mask) | (((ends >> (r * S::E)) & S::row_mask) << 1)) & ((2u << (unsigned) e) - 1) & S::row_mask);
				const offset_type into_piece = begins != 0 ? (offset_type) ((unsigned) e - (31 - __builtin_clz(begins))) : (offset_type) (at_row + e);
				const uint64_t segment = wave_segment + segments_past(wave_into_segment, offset_in_wave + (uint32_t) e, L);
				position_anchors[segment] = (uint_t<RunIndexSize>) piece;
				intra_run_anchor_offsets[segment] = (uint_t<RunLengthSize>) into_piece;
			}
		}
#ifdef GIDDY_CHAINED_STAMPS
		if (threadIdx.x == 0) { stamps[5] = __builtin_amdgcn_s_memrealtime(); }
#endif
	}
}

} // namespace one_pass

} // namespace runs
} // namespace compression
} // namespace kernels
} // namespace cuda
