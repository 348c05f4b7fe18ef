// The run encodings' compressors WITHOUT a chain (round 5): the column read once, no workgroup ever waits for another.
//
// The one-launch compressors of runs.hpp hand every tile the number of runs before it through a chain of tiles; a tile lives 26 microseconds, 8 of
// them reading the column — the rest waiting for the tiles before it and for the values it reads again —, and what a CU holds of such tiles moves
// 2.4 TB/s (run positions; run lengths, two chains: 1.7).  Here, every launch a grid of independent workgroups:
//
//   1. list_runs            a WAVEFRONT reads its 4096 elements (rows of 16 bytes per lane) and LISTS the runs that start among them — offset in the
//                           wavefront (16 bits) and value, out of the registers the comparison just used — in LDS, in element order (a row's ranks: a DPP
//                           scan for two rows), and copies the lists to the caller's scratch: `staged_per_wave` entries per wavefront at most (a wavefront of
//                           more runs leaves only their number: phase 3 lists it again).  The one launch that reads the column.
//   2. summarize_wavefronts a workgroup per BLOCK of 1024 wavefronts: the runs before each of them within the block, the block's number of runs; for run
//                           lengths instead: the first start after each wavefront and the last before it, within the block, and the block's first and last.
//      count_pieces         (run lengths) the same blocks: with the first start after every wavefront known — within its block, or a later block's first — the
//                           pieces of its runs (a wavefront's LAST run ends there, and needs more than one piece when it is longer than a length field
//                           holds), summed as the runs above.
//      place_blocks         ONE small workgroup: the runs (pieces) before every block; the grand total.
//   3. place_run_positions / place_run_lengths   a wavefront moves its lists to where its runs go, as positions or as lengths (the last run's pieces by
//                           the whole wavefront), and finds the anchors of the segments that begin among its elements by searching its list.  Twice: the
//                           wavefronts whose lists were staged (1 KiB of LDS each), and — a small grid at a stride — those of more runs than the scratch
//                           holds, listed again from the column (8 KiB of LDS each).
//
// Run lengths in 1-byte fields (pieces of 255 elements can end anywhere inside a wavefront's runs) stay with the chained kernel.
#pragma once
#include "giddy/kernels/compression/runs.hpp"

namespace cuda {
namespace kernels {
namespace compression {
namespace runs {
namespace listed {

using one_pass::per_lane;
using one_pass::wave_elements;
using one_pass::shape;

enum : unsigned {
	rows_in_flight = 8,              // rows (16 bytes per lane each) a wavefront of phase 1 asks for at a time (4 / 8 / 16 rows x 3 ... 8 wavefronts per SIMD: 190-200 us all, 2^28 elements)
	listing_waves_per_simd = 4,
	staged_per_wave_at_most = 512,   // entries of a wavefront's lists in LDS / in the scratch (a multiple of 16: the lists travel as 16-byte pieces)
	block_waves = 1024,              // wavefronts a workgroup of phase 2 takes: 4 per thread
	waves_per_thread = block_waves / threads_per_block,
	relisting_workgroups = 2048,     // the grid of phase 3's second launch (at a stride over the column's wavefronts)
};

/// what phase 1 leaves per wavefront: the number of runs that start among its elements | the offset of the first << 32 | of the last << 48
__host__ __device__ inline uint64_t header_of(uint32_t count, uint32_t first, uint32_t last) { return (uint64_t) count | ((uint64_t) (first & 0xFFFFu) << 32) | ((uint64_t) (last & 0xFFFFu) << 48); }
__host__ __device__ inline uint32_t count_in(uint64_t header) { return (uint32_t) header; }
__host__ __device__ inline uint32_t first_in(uint64_t header) { return (uint32_t) (header >> 32) & 0xFFFFu; }
__host__ __device__ inline uint32_t last_in(uint64_t header) { return (uint32_t) (header >> 48) & 0xFFFFu; }

/// The scratch of a column of `num_waves` wavefronts, cut up (uint64 entries; every part 16-byte aligned when the scratch is).
/// Per wavefront: headers, bases (runs or pieces before it WITHIN its block), next_starts / previous_starts (within its block; all ones / 0: none);
/// per block (`blocks` + i * blocks_stride, i = 0 ... 5): sums, bases, first_starts, last_starts (1 + a position; 0: none), next_starts, previous_starts.
struct scratch_layout {
	uint64_t headers, bases, previous_starts, next_starts, blocks, blocks_stride, num_blocks, offsets, values, entries;
	__host__ scratch_layout(uint64_t num_waves, unsigned element_size, unsigned staged_per_wave)
	{
		const uint64_t even = (num_waves + 1) / 2 * 2;
		num_blocks = (num_waves + block_waves - 1) / block_waves;
		blocks_stride = (num_blocks + 1) / 2 * 2;
		headers = 0; bases = even; previous_starts = 2 * even; next_starts = 3 * even; blocks = 4 * even;
		offsets = blocks + 6 * blocks_stride;
		values = offsets + (num_waves * staged_per_wave * 2 + 15) / 16 * 2;
		entries = values + (num_waves * staged_per_wave * element_size + 15) / 16 * 2;
	}
};

/// the most a wavefront may stage (a multiple of 16, at most staged_per_wave_at_most) so that the column's scratch fits `entries` uint64 words; 0: not even 16
__host__ inline unsigned staged_per_wave_that_fits(uint64_t entries, uint64_t num_waves, unsigned element_size)
{
	for (unsigned staged = staged_per_wave_at_most; staged >= 16; staged -= 16) {
		if (scratch_layout(num_waves, element_size, staged).entries <= entries) { return staged; }
	}
	return 0;
}

/// lane l's value in lane l + 1; lane 0: `into_first_lane` (one DPP move per dword: the whole wavefront shifted by a lane)
template <typename T>
__device__ __forceinline__ T from_the_lane_before(T value, T into_first_lane)
{
	enum : int { wave_shr_1 = 0x138 };
	if constexpr (sizeof(T) <= 4) {
		return (T) (uint32_t) __builtin_amdgcn_update_dpp((int) (uint32_t) into_first_lane, (int) (uint32_t) value, wave_shr_1, 0xf, 0xf, false);
	}
	else {
		const uint32_t low = (uint32_t) __builtin_amdgcn_update_dpp((int) (uint32_t) into_first_lane, (int) (uint32_t) value, wave_shr_1, 0xf, 0xf, false);
		const uint32_t high = (uint32_t) __builtin_amdgcn_update_dpp((int) (uint32_t) ((uint64_t) into_first_lane >> 32), (int) (uint32_t) ((uint64_t) value >> 32), wave_shr_1, 0xf, 0xf, false);
		return (T) (((uint64_t) high << 32) | low);
	}
}

/// element e of a row piece — 16 bytes as four dwords — (e: a constant after unrolling)
template <typename T>
__device__ __forceinline__ T element_at(const uint32_t (&words)[4], unsigned e)
{
	if constexpr (sizeof(T) == 8) { return (T) (((uint64_t) words[2 * e + 1] << 32) | words[2 * e]); }
	else {
		constexpr unsigned per_word = 4 / sizeof(T);
		uint32_t word = words[e / per_word];
		if constexpr (sizeof(T) < 4) { word = (word >> (8 * sizeof(T) * (e % per_word))) & ((1u << (8 * sizeof(T))) - 1u); }
		return (T) word;
	}
}

/// The runs that start among the wavefront's elements, in element order: entry k of `offsets` (and `values`) = the k-th such run's offset in the wavefront
/// (its value) for k < room; returns their number — which may be more than `room` — and the last one's offset.  The lists are LDS of this wavefront alone.
/// A row's starts are listed in as many turns as its busiest lane has of them (mean runs of 32: one or two — a branch per element was four to sixteen).
template <typename T, bool Whole, bool WithValues>
__device__ __forceinline__ uint32_t list_run_starts_of(const T* __restrict__ input, uint64_t wave_base, uint64_t n, unsigned lane,
	uint16_t* offsets, T* values, unsigned char* row_pad, uint32_t room, uint32_t& last_offset)
{
	using S = shape<sizeof(T)>;
	typedef uint32_t sixteen_bytes __attribute__((ext_vector_type(4)));
	// (rows in flight at a time; the column's last wavefront, or a column that is not 16-byte aligned: two rows, element by element)
	constexpr unsigned batch_rows = Whole ? (S::rows < rows_in_flight ? S::rows : (unsigned) rows_in_flight) : 2, batches = S::rows / batch_rows;
	static_assert(batch_rows % 2 == 0 && S::rows % batch_rows == 0, "rows are ranked two at a time");
	T carry = 0;   // what the first lane's first element of a row is compared with
	uint32_t running = 0, my_last = 0;   // runs in the rows before; 1 + the offset of the lane's last start
	#pragma unroll
	for (unsigned batch = 0; batch < batches; batch++) {
		uint32_t own[batch_rows][4];
		#pragma unroll
		for (unsigned i = 0; i < batch_rows; i++) {
			const uint64_t pos = wave_base + (batch * batch_rows + i) * S::row_elements + lane * S::E;
			if (Whole) {
				// (read once: non-temporal loads — alone, tools/read_probe.hip: 83-87 % of the HBM peak against 74-78 % with plain ones; this kernel: 190-198 us against 205-225)
				const sixteen_bytes loaded = __builtin_nontemporal_load(reinterpret_cast<const sixteen_bytes*>(input + pos));   // (read once: tools/read_probe.hip — 83-87 % of the HBM peak against 74-78 % with plain loads)
				own[i][0] = loaded.x; own[i][1] = loaded.y; own[i][2] = loaded.z; own[i][3] = loaded.w;
			}
			else {
				own[i][0] = own[i][1] = own[i][2] = own[i][3] = 0;
				#pragma unroll
				for (unsigned e = 0; e < S::E; e++) {
					const uint64_t element = pos + e < n ? (uint64_t) input[pos + e] : 0;
					if constexpr (sizeof(T) == 8) { own[i][2 * e] = (uint32_t) element; own[i][2 * e + 1] = (uint32_t) (element >> 32); }
					else { own[i][e * sizeof(T) / 4] |= (uint32_t) element << (8 * ((e * sizeof(T)) % 4)); }
				}
			}
		}
		if (batch == 0) { carry = wave_base == 0 ? (T) ~element_at<T>(own[0], 0) : input[wave_base - 1]; }
		#pragma unroll
		for (unsigned pair = 0; pair < batch_rows; pair += 2) {
			// two rows' starts, then ONE scan over the lanes for both (their counts — at most 1024 a row — in the halves of a dword)
			uint32_t bits[2];
			#pragma unroll
			for (unsigned h = 0; h < 2; h++) {
				const unsigned i = pair + h;
				const uint64_t pos = wave_base + (batch * batch_rows + i) * S::row_elements + lane * S::E;
				const T last_of_mine = element_at<T>(own[i], S::E - 1);
				T before = from_the_lane_before(last_of_mine, carry);
				bits[h] = 0;
				#pragma unroll
				for (unsigned e = 0; e < S::E; e++) {
					const T element = element_at<T>(own[i], e);
					bits[h] |= (uint32_t) ((Whole || pos + e < n) && element != before) << e;
					before = element;
				}
				carry = wave64::read_last_lane(last_of_mine);
			}
			const uint32_t mine = (uint32_t) __builtin_popcount(bits[0]) | ((uint32_t) __builtin_popcount(bits[1]) << 16);
			const uint32_t inclusive = wave64::inclusive_sum(mine), of_rows = wave64::read_last_lane(inclusive);
			const uint32_t exclusive = inclusive - mine;
			#pragma unroll
			for (unsigned h = 0; h < 2; h++) {
				const unsigned i = pair + h, r = batch * batch_rows + i;
				uint32_t rank = running + (h == 0 ? exclusive & 0xFFFFu : (of_rows & 0xFFFFu) + (exclusive >> 16));
				const uint32_t row_offset = r * S::row_elements + lane * S::E;
				// (the values of the starts: which of the lane's elements is a run-time matter — through 16 bytes of LDS of the lane's own, not through an array in memory)
				if (WithValues) { __builtin_memcpy(__builtin_assume_aligned(row_pad + 16 * lane, 16), own[i], 16); }   // (memcpy: bytes written as dwords and read as elements)
				for (uint32_t rest = bits[h]; wave64::ballot(rest != 0) != 0; ) {
					if (rest != 0) {
						const unsigned e = (unsigned) __builtin_ctz(rest);
						if (rank < room) {
							offsets[rank] = (uint16_t) (row_offset + e);
							if (WithValues) { T value; __builtin_memcpy(&value, __builtin_assume_aligned(row_pad + 16 * lane + sizeof(T) * e, sizeof(T)), sizeof(T)); values[rank] = value; }
						}
						rank++;
						rest &= rest - 1;
					}
				}
				my_last = bits[h] != 0 ? row_offset + (32u - (unsigned) __builtin_clz(bits[h])) : my_last;
			}
			running += (of_rows & 0xFFFFu) + (of_rows >> 16);
		}
		__builtin_amdgcn_sched_barrier(0);   // (the next batch's loads are not to be hoisted above this one's use: registers)
	}
	const uint32_t last = wave64::read_last_lane(wave64::inclusive_max(my_last));
	last_offset = last != 0 ? last - 1 : 0;
	wave64::lds_fence();
	return running;
}

template <typename T, bool WithValues>
__device__ __forceinline__ uint32_t list_run_starts(const T* __restrict__ input, uint64_t wave_base, uint64_t n, unsigned lane, uint16_t* offsets, T* values, unsigned char* row_pad,
	uint32_t room, uint32_t& last_offset)
{
	if (memory_access::is_aligned(input, 16) && wave_base + wave_elements <= n) { return list_run_starts_of<T, true, WithValues>(input, wave_base, n, lane, offsets, values, row_pad, room, last_offset); }
	return list_run_starts_of<T, false, WithValues>(input, wave_base, n, lane, offsets, values, row_pad, room, last_offset);
}

/// Phase 1.  `staged_per_wave`: a multiple of 16, at most staged_per_wave_at_most.
template <unsigned UncompressedSize>
__global__ void __launch_bounds__(threads_per_block, listing_waves_per_simd) list_runs(
	uint64_t*                        __restrict__  headers,
	uint16_t*                        __restrict__  staged_offsets,
	uint_t<UncompressedSize>*        __restrict__  staged_values,
	const uint_t<UncompressedSize>*  __restrict__  input,
	uint64_t                                       length,
	uint32_t                                       staged_per_wave)
{
	using T = uint_t<UncompressedSize>;
	typedef uint32_t sixteen_bytes __attribute__((ext_vector_type(4)));
	__shared__ alignas(16) uint16_t offsets[waves_per_block][staged_per_wave_at_most];
	__shared__ alignas(16) T values[waves_per_block][staged_per_wave_at_most];
	__shared__ alignas(16) unsigned char row_pads[waves_per_block][64 * 16];
	const unsigned lane = wave64::lane_id(), wave = wave64::index_in_block();
	const uint64_t n = length, num_waves = (n + wave_elements - 1) / wave_elements;
	for (uint64_t w = (uint64_t) blockIdx.x * waves_per_block + wave; w < num_waves; w += (uint64_t) gridDim.x * waves_per_block) {
		uint32_t last = 0;
		const uint32_t count = list_run_starts<T, true>(input, w * wave_elements, n, lane, offsets[wave], values[wave], row_pads[wave], staged_per_wave, last);
		if (lane == 0) { headers[w] = header_of(count, count != 0 ? offsets[wave][0] : 0, last); }
		if (count <= staged_per_wave) {
			// the lists leave as 16-byte pieces (what lies behind the last entry of a piece is never looked at)
			constexpr unsigned offsets_per_piece = 8, values_per_piece = 16 / UncompressedSize;
			for (uint32_t first = lane * offsets_per_piece; first < count; first += 64 * offsets_per_piece) {
				*reinterpret_cast<sixteen_bytes*>(staged_offsets + w * staged_per_wave + first) = *reinterpret_cast<const sixteen_bytes*>(offsets[wave] + first);
			}
			for (uint32_t first = lane * values_per_piece; first < count; first += 64 * values_per_piece) {
				*reinterpret_cast<sixteen_bytes*>(staged_values + w * staged_per_wave + first) = *reinterpret_cast<const sixteen_bytes*>(values[wave] + first);
			}
		}
		wave64::lds_fence();   // (the lists are rewritten by the wavefront's next 4096 elements)
	}
}

// ---- phase 2 -------------------------------------------------------------------------------------------------------------------------------------

struct minimum_of_uint64 {
	using result_type = uint64_t;
	__device__ static uint64_t neutral_value() { return ~(uint64_t) 0; }
	__device__ uint64_t operator()(uint64_t a, uint64_t b) const { return a < b ? a : b; }
};
struct maximum_of_uint64 {
	using result_type = uint64_t;
	__device__ static uint64_t neutral_value() { return 0; }
	__device__ uint64_t operator()(uint64_t a, uint64_t b) const { return a > b ? a : b; }
};
struct sum_of_uint64 {
	using result_type = uint64_t;
	__device__ static uint64_t neutral_value() { return 0; }
	__device__ uint64_t operator()(uint64_t a, uint64_t b) const { return a + b; }
};

/// what the threads before this one hold, combined (thread order), and what all of them do; `table`: an entry per wavefront of the workgroup (threads_per_block
/// threads).  Two barriers.
template <typename Op>
__device__ __forceinline__ uint64_t exclusive_over_the_workgroup(uint64_t mine, uint64_t* table, uint64_t& of_all)
{
	Op op;
	const uint64_t inclusive = wave64::inclusive_scan<Op>(mine);
	__syncthreads();   // (the table may still be read from an earlier call)
	if (wave64::lane_id() == 63) { table[wave64::index_in_block()] = inclusive; }
	__syncthreads();
	uint64_t before = Op::neutral_value(), all = Op::neutral_value();
	#pragma unroll
	for (unsigned w = 0; w < waves_per_block; w++) {
		const uint64_t t = table[w];
		if (w < wave64::index_in_block()) { before = op(before, t); }
		all = op(all, t);
	}
	of_all = all;
	return op(before, from_the_lane_before(inclusive, Op::neutral_value()));
}

/// how many pieces a run of `run_length` elements is stored as
__host__ __device__ inline uint64_t pieces_of(uint64_t run_length, uint64_t max_run_length) { return run_length <= max_run_length ? 1 : (run_length + max_run_length - 1) / max_run_length; }

/// where wavefront w's first / last run starts: all ones / 0 — none —, else the position / 1 + the position
__device__ __forceinline__ uint64_t first_start_of(uint64_t header, uint64_t w) { return count_in(header) != 0 ? w * wave_elements + first_in(header) : ~(uint64_t) 0; }
__device__ __forceinline__ uint64_t last_start_of(uint64_t header, uint64_t w) { return count_in(header) != 0 ? w * wave_elements + last_in(header) + 1 : 0; }

/// Phase 2, first launch: a workgroup per block of block_waves wavefronts.
/// Positions (!Lengths): bases[w] = the runs of the block's wavefronts before w; block_sums[b] = the block's runs.
/// Lengths: next_starts[w] / previous_starts[w] = the first start after / the last before wavefront w's elements WITHIN the block (all ones / 0: none; the latter
/// 1 + a position); block_first_starts[b] / block_last_starts[b] likewise of the whole block.
template <bool Lengths>
__global__ void __launch_bounds__(threads_per_block) summarize_wavefronts(
	uint64_t*        __restrict__  bases,
	uint64_t*        __restrict__  previous_starts,
	uint64_t*        __restrict__  next_starts,
	uint64_t*        __restrict__  block_sums,
	uint64_t*        __restrict__  block_first_starts,
	uint64_t*        __restrict__  block_last_starts,
	const uint64_t*  __restrict__  headers,
	uint64_t                       num_waves)
{
	__shared__ uint64_t table[waves_per_block];
	const uint64_t block_first_wave = (uint64_t) blockIdx.x * block_waves;
	if (!Lengths) {
		uint64_t counts[waves_per_thread], sum = 0;
		#pragma unroll
		for (unsigned q = 0; q < waves_per_thread; q++) {
			const uint64_t w = block_first_wave + threadIdx.x * waves_per_thread + q;
			counts[q] = w < num_waves ? (uint64_t) count_in(headers[w]) : 0;
			sum += counts[q];
		}
		uint64_t of_block;
		uint64_t before = exclusive_over_the_workgroup<sum_of_uint64>(sum, table, of_block);
		#pragma unroll
		for (unsigned q = 0; q < waves_per_thread; q++) {
			const uint64_t w = block_first_wave + threadIdx.x * waves_per_thread + q;
			if (w < num_waves) { bases[w] = before; }
			before += counts[q];
		}
		if (threadIdx.x == 0) { block_sums[blockIdx.x] = of_block; }
		return;
	}
	// the last start before every wavefront: in wavefront order
	{
		uint64_t last_start[waves_per_thread], furthest = 0;
		#pragma unroll
		for (unsigned q = 0; q < waves_per_thread; q++) {
			const uint64_t w = block_first_wave + threadIdx.x * waves_per_thread + q;
			last_start[q] = w < num_waves ? last_start_of(headers[w], w) : 0;
			furthest = furthest > last_start[q] ? furthest : last_start[q];
		}
		uint64_t of_block;
		uint64_t before = exclusive_over_the_workgroup<maximum_of_uint64>(furthest, table, of_block);
		#pragma unroll
		for (unsigned q = 0; q < waves_per_thread; q++) {
			const uint64_t w = block_first_wave + threadIdx.x * waves_per_thread + q;
			if (w < num_waves) { previous_starts[w] = before; }
			before = before > last_start[q] ? before : last_start[q];
		}
		if (threadIdx.x == 0) { block_last_starts[blockIdx.x] = of_block; }
	}
	// the first start after every wavefront: from the block's end — entry j of this order is the block's wavefront block_waves - 1 - j
	{
		uint64_t first_start[waves_per_thread], nearest = ~(uint64_t) 0;
		#pragma unroll
		for (unsigned q = 0; q < waves_per_thread; q++) {
			const uint64_t w = block_first_wave + (block_waves - 1 - (threadIdx.x * waves_per_thread + q));
			first_start[q] = w < num_waves ? first_start_of(headers[w], w) : ~(uint64_t) 0;
			nearest = nearest < first_start[q] ? nearest : first_start[q];
		}
		uint64_t of_block;
		uint64_t after = exclusive_over_the_workgroup<minimum_of_uint64>(nearest, table, of_block);
		#pragma unroll
		for (unsigned q = 0; q < waves_per_thread; q++) {
			const uint64_t w = block_first_wave + (block_waves - 1 - (threadIdx.x * waves_per_thread + q));
			if (w < num_waves) { next_starts[w] = after; }
			after = after < first_start[q] ? after : first_start[q];
		}
		if (threadIdx.x == 0) { block_first_starts[blockIdx.x] = of_block; }
	}
}

/// Phase 2, run lengths' second launch: the same blocks.  block_next_starts[b] / block_previous_starts[b] = the first start in a later block (the column's
/// length: none) / the last in an earlier one (1 + a position; 0: none); bases[w] = the PIECES of the block's wavefronts before w; block_sums[b] = the block's.
template <typename = void>   // (a template only so that the header may be included by several translation units)
__global__ void __launch_bounds__(threads_per_block) count_pieces(
	uint64_t*        __restrict__  bases,
	uint64_t*        __restrict__  block_sums,
	uint64_t*        __restrict__  block_next_starts,
	uint64_t*        __restrict__  block_previous_starts,
	const uint64_t*  __restrict__  next_starts,
	const uint64_t*  __restrict__  block_first_starts,
	const uint64_t*  __restrict__  block_last_starts,
	const uint64_t*  __restrict__  headers,
	uint64_t                       num_waves,
	uint64_t                       num_blocks,
	uint64_t                       length,
	uint64_t                       max_run_length)
{
	__shared__ uint64_t table[waves_per_block], second_table[waves_per_block];
	uint64_t nearest = length, furthest = 0;
	for (uint64_t b = (uint64_t) blockIdx.x + 1 + threadIdx.x; b < num_blocks; b += threads_per_block) { const uint64_t s = block_first_starts[b]; nearest = nearest < s ? nearest : s; }
	for (uint64_t b = threadIdx.x; b < blockIdx.x; b += threads_per_block) { const uint64_t s = block_last_starts[b]; furthest = furthest > s ? furthest : s; }
	uint64_t after_block, before_block;
	(void) exclusive_over_the_workgroup<minimum_of_uint64>(nearest, table, after_block);
	(void) exclusive_over_the_workgroup<maximum_of_uint64>(furthest, second_table, before_block);
	if (threadIdx.x == 0) { block_next_starts[blockIdx.x] = after_block; block_previous_starts[blockIdx.x] = before_block; }

	const uint64_t block_first_wave = (uint64_t) blockIdx.x * block_waves;
	uint64_t pieces[waves_per_thread], sum = 0;
	#pragma unroll
	for (unsigned q = 0; q < waves_per_thread; q++) {
		const uint64_t w = block_first_wave + threadIdx.x * waves_per_thread + q;
		pieces[q] = 0;
		if (w < num_waves) {
			const uint64_t header = headers[w], within = next_starts[w], after = within < after_block ? within : after_block;
			// the wavefront's runs but the last end inside it — shorter than any length field of two bytes or more holds
			if (count_in(header) != 0) { pieces[q] = count_in(header) - 1 + pieces_of(after - (w * wave_elements + last_in(header)), max_run_length); }
		}
		sum += pieces[q];
	}
	uint64_t of_block;
	uint64_t before = exclusive_over_the_workgroup<sum_of_uint64>(sum, table, of_block);
	#pragma unroll
	for (unsigned q = 0; q < waves_per_thread; q++) {
		const uint64_t w = block_first_wave + threadIdx.x * waves_per_thread + q;
		if (w < num_waves) { bases[w] = before; }
		before += pieces[q];
	}
	if (threadIdx.x == 0) { block_sums[blockIdx.x] = of_block; }
}

/// Phase 2, last launch — ONE workgroup: block_bases[b] = the runs (pieces) of the blocks before b; *grand_total = of all.
template <typename = void>
__global__ void __launch_bounds__(threads_per_block) place_blocks(uint64_t* __restrict__ block_bases, const uint64_t* __restrict__ block_sums, uint64_t num_blocks, uint64_t* __restrict__ grand_total)
{
	__shared__ uint64_t table[waves_per_block];
	uint64_t before_sweep = 0;
	for (uint64_t done = 0; done < num_blocks; done += threads_per_block) {   // (uniform trip count: the barriers inside are safe)
		const uint64_t b = done + threadIdx.x, mine = b < num_blocks ? block_sums[b] : 0;
		uint64_t of_sweep;
		const uint64_t before = before_sweep + exclusive_over_the_workgroup<sum_of_uint64>(mine, table, of_sweep);
		if (b < num_blocks) { block_bases[b] = before; }
		before_sweep += of_sweep;
	}
	if (threadIdx.x == 0 && grand_total != nullptr) { *grand_total = before_sweep; }
}

// ---- phase 3 -------------------------------------------------------------------------------------------------------------------------------------

/// how many of the list's `count` ascending offsets are at most `offset`
__device__ __forceinline__ uint32_t starts_up_to(const uint16_t* list, uint32_t count, uint32_t offset)
{
	uint32_t low = 0, high = count;   // entries below `low` are <= offset, entries from `high` on are not
	while (low < high) {
		const uint32_t middle = (low + high) / 2;
		if (list[middle] <= offset) { low = middle + 1; } else { high = middle; }
	}
	return low;
}

/// What both of phase 3's launches do with a wavefront — Relisted: false, the wavefronts whose lists were staged (the others are skipped); true, the others,
/// listed again from the column.  `emit(w, list, count, for_each_run)` does the scheme's part; for_each_run(f) calls f(i, value) for the lane's runs i of the list.
template <typename T, bool Relisted>
struct wavefront_lists {
	// (ONE launch for both kinds — 8 KiB of LDS a wavefront for all — was measured: 44-46 / 62 us against 31 + 6 / 45 + 6, run positions / lengths at 2^28 elements)
	enum : unsigned { list_entries = Relisted ? (unsigned) wave_elements : (unsigned) staged_per_wave_at_most, values_ahead = 4, offsets_ahead = 64 * values_ahead };
	typedef uint32_t sixteen_bytes __attribute__((ext_vector_type(4)));

	/// `prepare(w)`: what the scheme needs to know of wavefront w from the tables of phase 2 — asked for together with the header and the lists (scalar loads that are
	/// consumed one after the other are ISSUED one after the other, a round trip each: the first version's placing of run lengths made three of them behind the lists')
	template <typename Prepare, typename Emit>
	__device__ static __forceinline__ void for_each_wavefront(const T* __restrict__ input, const uint64_t* __restrict__ headers, const uint16_t* __restrict__ staged_offsets,
		const T* __restrict__ staged_values, uint32_t staged_per_wave, uint64_t n, Prepare&& prepare, Emit&& emit)
	{
		__shared__ alignas(16) uint16_t lists[waves_per_block][list_entries];
		const unsigned lane = wave64::lane_id(), wave = wave64::index_in_block();
		const uint64_t num_waves = (n + wave_elements - 1) / wave_elements;
		uint16_t* list = lists[wave];
		if constexpr (!Relisted) {
			for (uint64_t w = (uint64_t) blockIdx.x * waves_per_block + wave; w < num_waves; w += (uint64_t) gridDim.x * waves_per_block) {
				// the lists are asked for before the wavefront knows how long they are: one round trip for the header, the offsets and the first values
				const uint64_t header = headers[w];
				const auto placed = prepare(w);
				// (the first 256 of each: what a wavefront of mean runs of 16 or more has; a longer list's rest is asked for once its length is known)
				sixteen_bytes piece = { };
				if (lane * 8 < (staged_per_wave < offsets_ahead ? staged_per_wave : (uint32_t) offsets_ahead)) { piece = *reinterpret_cast<const sixteen_bytes*>(staged_offsets + w * staged_per_wave + lane * 8); }
				T ahead[values_ahead];
				#pragma unroll
				for (unsigned q = 0; q < values_ahead; q++) { ahead[q] = lane + 64 * q < staged_per_wave ? staged_values[w * staged_per_wave + lane + 64 * q] : (T) 0; }
				const uint32_t count = count_in(header);
				if (count > staged_per_wave) { continue; }   // (wave-uniform: the other launch's)
				{
					if (count > offsets_ahead && lane * 8 >= offsets_ahead && lane * 8 < count) { piece = *reinterpret_cast<const sixteen_bytes*>(staged_offsets + w * staged_per_wave + lane * 8); }
					*reinterpret_cast<sixteen_bytes*>(list + lane * 8) = piece;
					wave64::lds_fence();
					emit(w, placed, (const uint16_t*) list, count, [&](auto&& run) {
						#pragma unroll
						for (unsigned q = 0; q < values_ahead; q++) { if (lane + 64 * q < count) { run(lane + 64 * q, ahead[q]); } }
						for (uint32_t i = lane + 64 * values_ahead; i < count; i += 64) { run(i, staged_values[w * staged_per_wave + i]); }
					});
				}
				wave64::lds_fence();   // (the list is rewritten for the wavefront's next elements)
			}
		}
		else {
			// which wavefronts are this launch's: eight headers at a time (most columns have none: the launch is a look at the headers)
			enum : unsigned { looked_at_together = 8 };
			const uint64_t this_wave = (uint64_t) blockIdx.x * waves_per_block + wave, all_waves = (uint64_t) gridDim.x * waves_per_block;
			for (uint64_t first = this_wave * looked_at_together; first < num_waves; first += all_waves * looked_at_together) {
				const uint32_t of_lane = lane < looked_at_together && first + lane < num_waves ? count_in(headers[first + lane]) : 0;
				for (uint64_t todo = wave64::ballot(of_lane > staged_per_wave); todo != 0; todo &= todo - 1) {
					const unsigned which = (unsigned) __builtin_ctzll(todo);
					const uint64_t w = first + which;
					const uint32_t count = wave64::read_lane(of_lane, which);
					const auto placed = prepare(w);
					uint32_t last;
					(void) list_run_starts<T, false>(input, w * wave_elements, n, lane, list, (T*) nullptr, (unsigned char*) nullptr, (uint32_t) wave_elements, last);
					emit(w, placed, (const uint16_t*) list, count, [&](auto&& run) { for (uint32_t i = lane; i < count; i += 64) { run(i, input[w * wave_elements + list[i]]); } });
					wave64::lds_fence();   // (the list is rewritten for the next wavefront's elements)
				}
			}
		}
	}
};

/// Phase 3, run positions.
template <unsigned IndexSize, unsigned UncompressedSize, unsigned PositionSize, bool PositionsAreRelative, unsigned RunIndexSize, bool Relisted>
__global__ void __launch_bounds__(threads_per_block, one_pass::waves_per_simd) place_run_positions(
	uint_t<UncompressedSize>*        __restrict__  run_data,
	uint_t<PositionSize>*            __restrict__  run_start_positions,
	uint_t<RunIndexSize>*            __restrict__  position_anchors,
	const uint_t<UncompressedSize>*  __restrict__  input,
	const uint64_t*                  __restrict__  headers,
	const uint64_t*                  __restrict__  bases,
	const uint64_t*                  __restrict__  block_bases,
	const uint16_t*                  __restrict__  staged_offsets,
	const uint_t<UncompressedSize>*  __restrict__  staged_values,
	uint32_t                                       staged_per_wave,
	size_type_by_index_size<IndexSize>             segment_length,
	size_type_by_index_size<IndexSize>             length,
	uint64_t                                       capacity)
{
	using T = uint_t<UncompressedSize>;
	const unsigned lane = wave64::lane_id();
	const uint64_t n = length, L = segment_length;
	wavefront_lists<T, Relisted>::for_each_wavefront(input, headers, staged_offsets, staged_values, staged_per_wave, n,
		[&](uint64_t w) { const uint64_t within = bases[w], of_blocks = block_bases[w / block_waves]; return within + of_blocks; },
		[&](uint64_t w, uint64_t base, const uint16_t* list, uint32_t count, auto&& for_each_run) {
			const uint64_t wave_base = w * wave_elements;
			const uint64_t wave_into_segment = wave_base % L, wave_segment = wave_base / L;
			for_each_run([&](uint32_t i, T value) {
				const uint32_t offset = list[i];
				const uint64_t k = base + i;
				if (k < capacity) {
					run_data[k] = value;
					run_start_positions[k] = (uint_t<PositionSize>) (PositionsAreRelative ? one_pass::into_segment(wave_into_segment, offset, L) : wave_base + offset);
				}
			});
			// the anchors of the segments that begin among the wavefront's elements: the run covering a segment's first element is the last that starts at or before it
			const uint64_t wave_end = wave_base + wave_elements < n ? wave_base + wave_elements : n;
			const uint64_t first_segment = wave_into_segment == 0 ? wave_segment : wave_segment + 1;
			for (uint64_t s = first_segment + lane; s * L < wave_end; s += 64) {
				position_anchors[s] = (uint_t<RunIndexSize>) (base + starts_up_to(list, count, (uint32_t) (s * L - wave_base)) - 1);
			}
		});
}

/// Phase 3, run lengths in fields of two bytes or more (max_run_length >= 65 535: only a wavefront's LAST run can need more than one piece).
template <unsigned IndexSize, unsigned UncompressedSize, unsigned RunLengthSize, unsigned RunIndexSize, bool Relisted>
__global__ void __launch_bounds__(threads_per_block, one_pass::waves_per_simd) place_run_lengths(
	uint_t<UncompressedSize>*        __restrict__  run_data,
	uint_t<RunLengthSize>*           __restrict__  run_lengths,
	uint_t<RunIndexSize>*            __restrict__  position_anchors,
	uint_t<RunLengthSize>*           __restrict__  intra_run_anchor_offsets,
	const uint_t<UncompressedSize>*  __restrict__  input,
	const uint64_t*                  __restrict__  headers,
	const uint64_t*                  __restrict__  bases,
	const uint64_t*                  __restrict__  block_bases,
	const uint64_t*                  __restrict__  previous_starts,
	const uint64_t*                  __restrict__  next_starts,
	const uint64_t*                  __restrict__  block_previous_starts,
	const uint64_t*                  __restrict__  block_next_starts,
	const uint16_t*                  __restrict__  staged_offsets,
	const uint_t<UncompressedSize>*  __restrict__  staged_values,
	uint32_t                                       staged_per_wave,
	size_type_by_index_size<IndexSize>             segment_length,
	size_type_by_index_size<IndexSize>             length,
	uint64_t                                       max_run_length,
	uint64_t                                       capacity)
{
	static_assert(RunLengthSize >= 2, "runs in 1-byte length fields: the chained kernel");
	using T = uint_t<UncompressedSize>;
	const unsigned lane = wave64::lane_id();
	const uint64_t n = length, L = segment_length, M = max_run_length;
	struct placed_wavefront { uint64_t base, next_start, previous_start; };
	wavefront_lists<T, Relisted>::for_each_wavefront(input, headers, staged_offsets, staged_values, staged_per_wave, n,
		[&](uint64_t w) {
			const uint64_t block = w / block_waves;
			// (all six asked for before any is used)
			const uint64_t within = bases[w], of_blocks = block_bases[block], next_within = next_starts[w], next_beyond = block_next_starts[block];
			const uint64_t previous_within = previous_starts[w], previous_beyond = block_previous_starts[block];
			const uint64_t previous_plus_one = previous_within > previous_beyond ? previous_within : previous_beyond;
			return placed_wavefront { within + of_blocks, next_within < next_beyond ? next_within : next_beyond, previous_plus_one != 0 ? previous_plus_one - 1 : 0 };
		},
		[&](uint64_t w, const placed_wavefront& placed, const uint16_t* list, uint32_t count, auto&& for_each_run) {
			const uint64_t wave_base = w * wave_elements, base = placed.base, next_start = placed.next_start, previous_start = placed.previous_start;
			for_each_run([&](uint32_t i, T value) {
				const uint32_t offset = list[i];
				const uint64_t k = base + i;
				// a run that ends inside the wavefront: one piece; the last one ends where the next run starts, wavefronts on maybe: its first piece
				const uint64_t run_length = i + 1 < count ? (uint64_t) (list[i + 1] - offset) : next_start - (wave_base + offset);
				if (k < capacity) { run_data[k] = value; run_lengths[k] = (uint_t<RunLengthSize>) (run_length > M ? M : run_length); }
			});
			if (count != 0) {   // ... and the last one's pieces beyond the first, if any, by all lanes
				const uint32_t offset = list[count - 1];
				const uint64_t run_length = next_start - (wave_base + offset), pieces = pieces_of(run_length, M);
				if (pieces > 1) {
					const T value = input[wave_base + offset];
					for (uint64_t j = 1 + lane; j < pieces; j += 64) {
						const uint64_t k = base + count - 1 + j;
						if (k < capacity) { run_data[k] = value; run_lengths[k] = (uint_t<RunLengthSize>) (j + 1 < pieces ? M : run_length - (pieces - 1) * M); }
					}
				}
			}
			// the anchors of the segments that begin among the wavefront's elements: the piece holding the segment's first element, and how far into it
			const uint64_t wave_end = wave_base + wave_elements < n ? wave_base + wave_elements : n;
			const uint64_t first_segment = (wave_base + L - 1) / L;
			for (uint64_t s = first_segment + lane; s * L < wave_end; s += 64) {
				const uint64_t x = s * L;
				const uint32_t up_to = starts_up_to(list, count, (uint32_t) (x - wave_base));
				uint64_t piece, into_piece;
				if (up_to != 0) {   // a run of this wavefront's: the element is less than a wavefront's elements into it — in its first piece
					piece = base + up_to - 1;
					into_piece = x - (wave_base + list[up_to - 1]);
				}
				else {   // the run that reaches into the wavefront: its pieces are the last before the wavefront's own
					const uint64_t ends = count != 0 ? wave_base + list[0] : next_start;
					const uint64_t into_run = x - previous_start;
					piece = base - pieces_of(ends - previous_start, M) + into_run / M;
					into_piece = into_run % M;
				}
				position_anchors[s] = (uint_t<RunIndexSize>) piece;
				intra_run_anchor_offsets[s] = (uint_t<RunLengthSize>) into_piece;
			}
		});
}

} // namespace listed
} // namespace runs
} // namespace compression
} // namespace kernels
} // namespace cuda
