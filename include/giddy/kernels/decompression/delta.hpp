// Delta decompression for CDNA4:
//
//   out[s*L + i] = anchor_values[s] + sum_{j <= i} compressed_input[s*L + j]      (mod 2^(8*UncompressedSize))
//
// i.e. a per-segment inclusive prefix sum of UNSIGNED deltas seeded with the segment's anchor —
// the element semantics of the reference's src/kernels/decompression/delta.cuh:66-108 (+
// src/kernels/reduction/scan.cuh:285-381), same kernel template name and parameter list.
//
// The reference scans one element per thread per block-wide chunk: a 5-step shuffle scan, a
// shared-memory exchange, two __syncthreads and a second scan by warp 0 for every blockDim
// elements — synchronisation-bound, and only correct when segment_length >= blockDim (it indexes
// anchors by a block-sized pseudo-segment, delta.cuh:80-92).  Here:
//
//  * rows.  A wavefront works on rows of 64*E consecutive elements, E = 16/UncompressedSize, lane l
//    owning elements [l*E, l*E+E): one 2..16-byte load of E deltas, E-1 adds in registers, a
//    wave-wide DPP inclusive sum of the 64 lane totals (6 VALU steps, no LDS), one 16-byte store
//    that is contiguous across the wavefront.  R rows are in flight per wavefront; their scans are
//    independent and only the R row totals chain.  When the deltas are <= 2 bytes wide the
//    in-row scan runs in 32 bits even for 64-bit outputs (a row's prefix cannot overflow).
//  * wave path (segment_length % row == 0, at most 32 rows = 32 KiB of output per segment): one
//    wavefront per anchored segment — or per batch of short segments adding up to 32 rows — carry in
//    a register, no barriers; the next group's loads are issued before this group's stores.  (Rows in
//    flight were measured per instantiation — 1, 2 or 8, rows_in_flight_for below; at 4096-element
//    segments of 64-bit values the block path is no faster, from 8192 elements on it is.)
//  * staged wave path (a lane's deltas of a row are at most 4 bytes: <8,1>, <8,2>, <4,1>): the chunk's whole
//    input — at most 8 KiB — arrives first as 16-byte-per-lane loads and is handed out row by row through
//    wave-private LDS, so the row loop issues stores only.  C3 (uint64 <- uint16): 74 -> 79 % of HBM peak.
//  * block path (longer segments): a workgroup of ceil(rows per segment / 8) wavefronts (at most 16)
//    owns a segment; wavefront w takes 8 consecutive rows of each sweep, all its loads are issued
//    before the single barrier of the sweep, carries cross wavefronts through a double-buffered
//    16-entry LDS table: one barrier per waves*8 rows.
//  * ragged wave path (segment_length a multiple of E but not of a row, e.g. 1000): the same wave path with rows laid
//    out from each segment's start and the last row of a segment cut short; the block path takes any such length as is.
//  * general path (segment_length not a multiple of E, or unaligned pointers): a segmented scan
//    (reset flags at segment heads) over rows, anchors gathered per element.  Slower, correct for
//    every L — including L < blockDim, which the reference gets wrong.
//
// HBM traffic per element: CompressedSize read + UncompressedSize written (+ one anchor per
// segment): 10.002 B for 64-bit <- 16-bit at L = 4096.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <stdexcept>

#include "giddy/common.hpp"
#include "giddy/kernels/launch_config_resolution_params.hpp"
#include "giddy/on_device/batch_pieces.hpp"
#include "giddy/on_device/sorted_patches.hpp"
#include "giddy/kernels/tuning.hpp"
#include "giddy/on_device/wave64.hpp"
#include "giddy/util/math.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace delta {

namespace detail {

// What was measured on MI355X and is no longer a build switch (the experiments' numbers are in profiles/r0*_delta_*.txt):
//   * segments that are not whole rows go by rows laid out on the OUTPUT's cache lines, a segment boundary inside a row (wave_per_ragged_segment_in_aligned_rows;
//     rows that begin where the segment does leave partial lines at both ends of every row);
//   * rows in flight per wavefront are per instantiation (rows_in_flight_for) — there is no single best value;
//   * the whole-row path is not the ragged path with nothing ragged (its bookkeeping costs 2-4 points).
/// bytes of input per lane per row up to which a chunk's input goes through LDS on the wave path — and in the fused kernels, whose patch windows share the LDS
enum : unsigned { stage_row_inputs_up_to = 4, patched_stage_row_inputs_up_to = 2 };
/// ragged segments: staging their input pays for 64-bit elements (<4,8,2>: 69 % of the HBM peak against 60; <4,4,1>: 76 % against 80)
constexpr bool ragged_segments_staged(unsigned uncompressed_size) { return uncompressed_size == 8; }

enum : unsigned {
	rows_per_wave_per_sweep = 8,                                                  // block path
	wave_path_threads = 256,
	rows_per_chunk = 32,                                                                                  // wave path: rows a wavefront takes per outer-loop trip;
	                                                                              // segments of more rows are swept by a whole workgroup
	max_waves_per_block = 16,
};

/// Rows a wavefront of the wave path keeps in flight.  Measured on MI355X at 2^30 elements, segments of 4096, with 1, 2, 4
/// and 8 rows (% of the 8 TB/s peak): <8,1>: 63 57 55 51; <8,2>: 70 75 73 74; <8,4>: 74 76 73 68; <4,1>: 70 75 73 72;
/// <4,2>: 77 72 72 72; <2,1>: 65 67 74 77 — there is no single best value, so each instantiation gets its own.
template <unsigned UncompressedSize, unsigned CompressedSize>
constexpr unsigned rows_in_flight_for()
{
	if (UncompressedSize == 2) { return 8; }
	if ((UncompressedSize == 8 && CompressedSize == 1) || (UncompressedSize == 4 && CompressedSize == 2)) { return 1; }
	return 2;
}

template <unsigned UncompressedSize, unsigned CompressedSize>
struct row_traits {
	enum : unsigned { E = 16 / UncompressedSize, row = 64 * E, rows_in_flight = rows_in_flight_for<UncompressedSize, CompressedSize>() };
	using uncompressed_type = uint_t<UncompressedSize>;
	using compressed_type = uint_t<CompressedSize>;
	// type the in-row prefix is formed in: 64*E elements of at most 2^(8*CompressedSize)-1 each
	using scan_type = typename std::conditional<(UncompressedSize <= 4 || CompressedSize <= 2), uint32_t, uint64_t>::type;
};

/// One row: load, scan, return the exclusive-of-row prefixes; `valid` = elements of the row inside the input
template <unsigned UncompressedSize, unsigned CompressedSize>
struct row_t {
	using traits = row_traits<UncompressedSize, CompressedSize>;
	using scan_type = typename traits::scan_type;
	enum : unsigned { E = traits::E };
	memory_access::group<typename traits::compressed_type, E> deltas;
	scan_type prefix[E];     // inclusive prefix within the row, per owned element
	scan_type total;         // sum of the whole row (uniform)

	template <bool AnyAlignment = false>   // (the group's address aligned for an element only: decompress_at_any_alignment)
	__device__ __forceinline__ void load(const typename traits::compressed_type* __restrict__ in, uint64_t row_start, uint64_t end, unsigned lane)
	{
		uint64_t p = row_start + (uint64_t) lane * E;
		if (p + E <= end) {
			if constexpr (AnyAlignment) { deltas = memory_access::load_group_at_any_alignment<typename traits::compressed_type, E>(in + p); }
			else { deltas = memory_access::load_group<typename traits::compressed_type, E>(in + p); }
		}
		else {
			#pragma unroll
			for (unsigned j = 0; j < E; j++) { deltas[j] = (p + j < end) ? in[p + j] : (typename traits::compressed_type) 0; }
		}
	}
	__device__ __forceinline__ void scan()
	{
		scan_type running = 0;
		#pragma unroll
		for (unsigned j = 0; j < E; j++) { running += deltas[j]; prefix[j] = running; }
		scan_type inclusive = wave64::inclusive_sum(running);
		scan_type exclusive = inclusive - running;
		#pragma unroll
		for (unsigned j = 0; j < E; j++) { prefix[j] += exclusive; }
		total = wave64::read_last_lane(inclusive);
	}
	template <typename Patches>
	__device__ __forceinline__ void store(typename traits::uncompressed_type* __restrict__ out, uint64_t row_start, uint64_t end, unsigned lane,
		typename traits::uncompressed_type base, Patches& patches) const
	{
		using U = typename traits::uncompressed_type;
		uint64_t p = row_start + (uint64_t) lane * E;
		memory_access::group<U, E> values;
		#pragma unroll
		for (unsigned j = 0; j < E; j++) { values[j] = (U) (base + (U) prefix[j]); }
		if (row_start < end) { patches.template apply_to_row<E>(row_start, (row_start + 64 * E < end) ? row_start + 64 * E : end, lane, values); }   // (rows past the end: nothing to patch, and no walking into the next segment's patches)
		if (p + E <= end) { memory_access::store_group<memory_access::wide_output_policy<sizeof(U)>(GIDDY_DELTA_STORE_POLICY)>(out + p, values); }
		else {
			#pragma unroll
			for (unsigned j = 0; j < E; j++) { if (p + j < end) { out[p + j] = values[j]; } }
		}
	}
	/// ... with the values worked out by the caller (every element its own base)
	template <typename Patches>
	__device__ __forceinline__ void store_values(typename traits::uncompressed_type* __restrict__ out, uint64_t row_start, uint64_t end, unsigned lane,
		memory_access::group<typename traits::uncompressed_type, E> values, Patches& patches) const
	{
		using U = typename traits::uncompressed_type;
		uint64_t p = row_start + (uint64_t) lane * E;
		if (row_start < end) { patches.template apply_to_row<E>(row_start, (row_start + 64 * E < end) ? row_start + 64 * E : end, lane, values); }
		if (p + E <= end) { memory_access::store_group<memory_access::wide_output_policy<sizeof(U)>(GIDDY_DELTA_STORE_POLICY)>(out + p, values); }
		else {
			#pragma unroll
			for (unsigned j = 0; j < E; j++) { if (p + j < end) { out[p + j] = values[j]; } }
		}
	}
};

/// Segments per wavefront chunk on the wave path: short segments are batched so that a wavefront always has
/// about `rows_per_chunk` rows of work between two trips around its outer loop (kernel and launch parameters agree).
__host__ __device__ inline uint64_t segments_per_wave_chunk(uint64_t rows_per_segment)
{
	return rows_per_segment >= rows_per_chunk ? 1 : rows_per_chunk / rows_per_segment;
}

/// Whether the segments of a column go a WAVEFRONT each (kernel and launch parameters agree): those of up to `rows_per_chunk` rows — and,
/// round 4, longer RAGGED segments of 64-bit elements when there are thousands of them: the workgroup path's rows begin where the segment does
/// (partial cache lines at both ends of every row) and its input loads are 2-8 bytes per lane — 46-68 % of the HBM peak at segments of 5000
/// and 10000 elements where the wave paths for ragged segments (rows by the output's cache lines, input through LDS) reach 58-74 %.  For
/// 2- and 4-byte elements the workgroup path is the faster one (78-89 %), and for 2-byte deltas of 64-bit elements too (68 % against 64-67).
enum : unsigned { long_ragged_by_wave_from_segments = 8192 };   // (segments of more than rows_per_chunk rows, ragged, 64-bit elements: a wavefront each from this many segments on)
template <unsigned UncompressedSize, unsigned CompressedSize>
__host__ __device__ inline bool takes_wave_path(uint64_t n, uint64_t L)
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	const uint64_t rows = (L + traits::row - 1) / traits::row;
	// (Round 4, measured and NOT adopted: whole-row segments of 8-32 rows of 4- and 2-byte elements from 1-byte deltas by WORKGROUPS of four wavefronts, 8 rows apiece, all
	// eight in flight — a segment of 8 or 16 rows sharing its workgroup with three or one more (block_per_segment).  At 2^28 elements: <4,4,1> at segments of 2048 / 4096 /
	// 8192: 82 / 82 / 83 -> 89 % of the HBM peak, <4,2,1>: 82 -> 86 %.  At 2^30: 73 / 73 / 71 % against 73 / 78 / 77 % — the 2^28 column's 256 MiB of deltas sit in the
	// Infinity Cache between repetitions, and the workgroup shape profits from that more than it costs; read from HBM it is the slower one.)
	if (rows <= rows_per_chunk) { return true; }
	return UncompressedSize == 8 && CompressedSize != 2 && L % traits::row != 0 && n / long_ragged_by_wave_from_segments >= L;
}

/// A chunk's anchors.  (Tried in round 4 and taken out: ONE vector load of all of a chunk's anchors, a lane each, ahead of the chunk's first rows, instead of a load at every
/// segment's first row — 2-byte elements -5 points, <8,8,2> -1.5, the rest within noise.)
template <typename U>
struct chunk_anchors {
	const U* __restrict__ anchors;
	__device__ __forceinline__ void load(const U* __restrict__ anchors_, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t, unsigned) { anchors = anchors_; }
	__device__ __forceinline__ U of(uint64_t segment) const { return anchors[segment]; }
};

/// wave path: this wavefront decompresses whole segments on its own (several short ones per chunk)
template <unsigned UncompressedSize, unsigned CompressedSize, typename Patches, typename Grid = launch_grid::whole>
__device__ __forceinline__ void wave_per_segment(
	uint_t<UncompressedSize>* __restrict__ out, const uint_t<CompressedSize>* __restrict__ in,
	const uint_t<UncompressedSize>* __restrict__ anchors, uint64_t n, uint64_t L, Patches patches,
	Grid grid = Grid())
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	using U = typename traits::uncompressed_type;
	constexpr unsigned R = traits::rows_in_flight;
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) grid.block() * wave64::per_block() + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) grid.blocks() * wave64::per_block();
	const uint64_t segments_per_chunk = segments_per_wave_chunk(L / traits::row);
	const uint64_t chunk_length = segments_per_chunk * L;
	const uint64_t num_chunks = (n + chunk_length - 1) / chunk_length;

	for (uint64_t c = wave; c < num_chunks; c += num_waves) {
		const uint64_t start = c * chunk_length;
		const uint64_t end = (start + chunk_length < n) ? start + chunk_length : n;
		uint64_t segment = c * segments_per_chunk, next_segment_start = start;   // the chunk starts on a segment boundary
		chunk_anchors<U> chunk_anchor_values;
		chunk_anchor_values.load(anchors, segment, segments_per_chunk, start, end, L, lane);
		U base = 0;
		// Software-pipelined: the NEXT group's loads are issued before this group's stores.  CDNA4 retires
		// vector-memory operations in order, so a load issued after a store cannot be waited for without
		// waiting for the store too; this way a wavefront waits for one HBM round trip per group, not two.
		row_t<UncompressedSize, CompressedSize> rows[R], next[R];
		patches.begin_seek(start, lane);   // (its probes go out ahead of the first rows' loads, the rest behind them)
		#pragma unroll
		for (unsigned k = 0; k < R; k++) { rows[k].load(in, start + (uint64_t) k * traits::row, end, lane); }
		patches.finish_seek(lane);
		for (uint64_t row_start = start; row_start < end; row_start += (uint64_t) R * traits::row) {
			const uint64_t next_start = row_start + (uint64_t) R * traits::row;
			if (next_start < end) {
				#pragma unroll
				for (unsigned k = 0; k < R; k++) { next[k].load(in, next_start + (uint64_t) k * traits::row, end, lane); }
			}
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { rows[k].scan(); }
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				const uint64_t this_row = row_start + (uint64_t) k * traits::row;
				if (this_row == next_segment_start && this_row < end) { base = chunk_anchor_values.of(segment++); next_segment_start += L; }   // wave-uniform
				rows[k].store(out, this_row, end, lane, base, patches);
				base += (U) rows[k].total;
			}
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { rows[k].deltas = next[k].deltas; }
		}
	}
}

/// wave path for 1-byte deltas of 8-byte elements, where a row's input is only 128 bytes: the chunk's whole input
/// (at most 4 KiB) arrives first, as 16-byte-per-lane loads, and is handed out row by row through wave-private LDS.
/// (Measured with the bare access pattern, round 4's stream probe — docs/history/r04.md: 68 % of peak with 2-byte loads per row, 73 % with
/// wide loads through LDS; the decoder itself: 63 -> see DESIGN.md.)
template <unsigned UncompressedSize, unsigned CompressedSize, typename Patches, typename Grid = launch_grid::whole>
__device__ __forceinline__ void wave_per_segment_with_staged_input(
	uint_t<UncompressedSize>* __restrict__ out, const uint_t<CompressedSize>* __restrict__ in,
	const uint_t<UncompressedSize>* __restrict__ anchors, uint64_t n, uint64_t L, Patches patches, uint8_t* staging,
	Grid grid = Grid())
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	using U = typename traits::uncompressed_type;
	using C = typename traits::compressed_type;
	typedef uint32_t piece_t __attribute__((ext_vector_type(4)));
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) grid.block() * wave64::per_block() + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) grid.blocks() * wave64::per_block();
	const uint64_t segments_per_chunk = segments_per_wave_chunk(L / traits::row);
	const uint64_t chunk_length = segments_per_chunk * L;
	const uint64_t num_chunks = (n + chunk_length - 1) / chunk_length;
	const uint8_t* in_bytes = reinterpret_cast<const uint8_t*>(in);

	for (uint64_t c = wave; c < num_chunks; c += num_waves) {
		const uint64_t start = c * chunk_length;
		const uint64_t end = (start + chunk_length < n) ? start + chunk_length : n;
		const unsigned chunk_bytes = (unsigned) (end - start) * CompressedSize;
		chunk_anchors<U> chunk_anchor_values;
		chunk_anchor_values.load(anchors, c * segments_per_chunk, segments_per_chunk, start, end, L, lane);
		patches.begin_seek(start, lane);
		wave64::lds_fence();   // (the previous chunk's rows have been read)
		for (unsigned at = lane * 16; at < chunk_bytes; at += 64 * 16) {
			if (at + 16 <= chunk_bytes) {
				*reinterpret_cast<piece_t*>(staging + at) = *reinterpret_cast<const piece_t*>(in_bytes + start * CompressedSize + at);
			}
			else {
				for (unsigned b = at; b < chunk_bytes; b++) { staging[b] = in_bytes[start * CompressedSize + b]; }
			}
		}
		patches.finish_seek(lane);
		wave64::lds_fence();
		uint64_t segment = c * segments_per_chunk, next_segment_start = start;   // the chunk starts on a segment boundary
		U base = 0;
		for (uint64_t row_start = start; row_start < end; row_start += traits::row) {
			row_t<UncompressedSize, CompressedSize> row;
			const uint64_t p = row_start + (uint64_t) lane * traits::E;
			const C* staged = reinterpret_cast<const C*>(staging + (p - start) * CompressedSize);
			#pragma unroll
			for (unsigned j = 0; j < traits::E; j++) { row.deltas[j] = (p + j < end) ? staged[j] : (C) 0; }
			row.scan();
			if (row_start == next_segment_start) { base = chunk_anchor_values.of(segment++); next_segment_start += L; }   // wave-uniform
			row.store(out, row_start, end, lane, base, patches);
			base += (U) row.total;
		}
	}
}

/// wave path for segments that are not whole rows (segment_length % E == 0 is all it takes for 16-byte access:
/// 1000, 3000, 10000 ...): rows are laid out from each segment's start, the last row of a segment is cut short,
/// and the next segment's first row begins right where it ends — same pipeline, rows of two lengths.
template <unsigned UncompressedSize, unsigned CompressedSize, typename Patches, typename Grid = launch_grid::whole>
__device__ __forceinline__ void wave_per_ragged_segment(
	uint_t<UncompressedSize>* __restrict__ out, const uint_t<CompressedSize>* __restrict__ in,
	const uint_t<UncompressedSize>* __restrict__ anchors, uint64_t n, uint64_t L, Patches patches,
	Grid grid = Grid())
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	using U = typename traits::uncompressed_type;
	constexpr unsigned R = traits::rows_in_flight;
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) grid.block() * wave64::per_block() + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) grid.blocks() * wave64::per_block();
	const uint64_t segments_per_chunk = segments_per_wave_chunk((L + traits::row - 1) / traits::row);
	const uint64_t chunk_length = segments_per_chunk * L;
	const uint64_t num_chunks = (n + chunk_length - 1) / chunk_length;

	struct row_bounds { uint64_t start, end; bool first_of_segment; };   // wave-uniform
	for (uint64_t c = wave; c < num_chunks; c += num_waves) {
		const uint64_t chunk_start = c * chunk_length;
		const uint64_t chunk_end = (chunk_start + chunk_length < n) ? chunk_start + chunk_length : n;
		uint64_t segment = c * segments_per_chunk;
		// the walk over the chunk's rows: [start, end) with end = start + row, or the segment's end if that comes first
		uint64_t walk_start = chunk_start;
		uint64_t walk_segment_end = (chunk_start + L < chunk_end) ? chunk_start + L : chunk_end;
		bool walk_first = true;
		auto take_row = [&]() {
			row_bounds r;
			r.start = walk_start; r.first_of_segment = walk_first;
			r.end = (walk_start + traits::row < walk_segment_end) ? walk_start + traits::row : walk_segment_end;
			walk_start = r.end;
			walk_first = walk_start == walk_segment_end;
			if (walk_first) { walk_segment_end = (walk_segment_end + L < chunk_end) ? walk_segment_end + L : chunk_end; }
			return r;   // past the chunk's end: start == end == chunk_end, an empty row
		};
		U base = 0;
		chunk_anchors<U> chunk_anchor_values;
		chunk_anchor_values.load(anchors, segment, segments_per_chunk, chunk_start, chunk_end, L, lane);
		row_t<UncompressedSize, CompressedSize> rows[R], next[R];
		row_bounds bounds[R], next_bounds[R];
		patches.begin_seek(chunk_start, lane);
		#pragma unroll
		for (unsigned k = 0; k < R; k++) { bounds[k] = take_row(); rows[k].load(in, bounds[k].start, bounds[k].end, lane); }
		patches.finish_seek(lane);
		while (bounds[0].start < chunk_end) {
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { next_bounds[k] = take_row(); next[k].load(in, next_bounds[k].start, next_bounds[k].end, lane); }
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { rows[k].scan(); }
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				if (bounds[k].start >= chunk_end) { continue; }   // wave-uniform
				if (bounds[k].first_of_segment) { base = chunk_anchor_values.of(segment++); }
				rows[k].store(out, bounds[k].start, bounds[k].end, lane, base, patches);
				base += (U) rows[k].total;
			}
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { rows[k].deltas = next[k].deltas; bounds[k] = next_bounds[k]; }
		}
	}
}

/// Ragged segments, rows laid out by the OUTPUT (round 4).  wave_per_ragged_segment above starts a row wherever a segment starts: with segments
/// of 1000 or 3000 64-bit elements no row begins on a cache line, every 1 KiB row store is nine partial lines instead of eight whole ones, and
/// the decoder runs at 61 % of the HBM peak where whole-row segments reach 72-78 %.  Here the chunk's rows begin on ROW-sized boundaries of the output
/// (a first partial row, then whole aligned rows, a last partial one) whatever the segments do: a row may have ONE segment boundary inside it
/// (segments of at least a row) — lanes are whole groups of E elements on either side, since segment lengths are multiples of E — and the lanes
/// behind it count from the new segment's anchor: value = anchor' + (prefix - the prefix at the boundary).  Same pipeline: the next rows' loads
/// before these rows' stores.
template <unsigned UncompressedSize, unsigned CompressedSize, typename Patches, typename Grid = launch_grid::whole>
__device__ __forceinline__ void wave_per_ragged_segment_in_aligned_rows(
	uint_t<UncompressedSize>* __restrict__ out, const uint_t<CompressedSize>* __restrict__ in,
	const uint_t<UncompressedSize>* __restrict__ anchors, uint64_t n, uint64_t L, Patches patches,
	Grid grid = Grid())
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	using U = typename traits::uncompressed_type;
	using scan_type = typename traits::scan_type;
	constexpr unsigned R = traits::rows_in_flight > 4 ? 4 : traits::rows_in_flight;   // (8 rows of 2-byte elements with their bounds in flight spill registers)
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) grid.block() * wave64::per_block() + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) grid.blocks() * wave64::per_block();
	const uint64_t segments_per_chunk = segments_per_wave_chunk((L + traits::row - 1) / traits::row);
	const uint64_t chunk_length = segments_per_chunk * L;
	const uint64_t num_chunks = (n + chunk_length - 1) / chunk_length;

	struct row_bounds { uint64_t start, end; };   // wave-uniform
	for (uint64_t c = wave; c < num_chunks; c += num_waves) {
		const uint64_t chunk_start = c * chunk_length;
		const uint64_t chunk_end = (chunk_start + chunk_length < n) ? chunk_start + chunk_length : n;
		// elements from the chunk's start to the next row-sized boundary of the OUTPUT (a multiple of E: the output is 16-byte aligned)
		const uint64_t misplaced = (reinterpret_cast<uintptr_t>(out + chunk_start) / sizeof(U)) % traits::row;
		const uint64_t head = misplaced == 0 ? traits::row : traits::row - misplaced;
		uint64_t walk_start = chunk_start;
		auto take_row = [&]() {
			row_bounds r;
			r.start = walk_start;
			const uint64_t length = walk_start == chunk_start ? head : traits::row;
			r.end = walk_start + length < chunk_end ? walk_start + length : chunk_end;
			walk_start = r.end;
			return r;   // past the chunk's end: start == end == chunk_end, an empty row
		};
		uint64_t segment = c * segments_per_chunk, next_boundary = chunk_start;   // the next segment's first element
		U base = 0;
		chunk_anchors<U> chunk_anchor_values;
		chunk_anchor_values.load(anchors, segment, segments_per_chunk, chunk_start, chunk_end, L, lane);
		row_t<UncompressedSize, CompressedSize> rows[R], next[R];
		row_bounds bounds[R], next_bounds[R];
		patches.begin_seek(chunk_start, lane);
		#pragma unroll
		for (unsigned k = 0; k < R; k++) { bounds[k] = take_row(); rows[k].load(in, bounds[k].start, bounds[k].end, lane); }
		patches.finish_seek(lane);
		while (bounds[0].start < chunk_end) {
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { next_bounds[k] = take_row(); next[k].load(in, next_bounds[k].start, next_bounds[k].end, lane); }
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { rows[k].scan(); }
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				if (bounds[k].start >= chunk_end) { continue; }   // wave-uniform
				if (bounds[k].start == next_boundary) { base = chunk_anchor_values.of(segment++); next_boundary += L; }
				if (next_boundary < bounds[k].end) {
					// a segment begins inside the row, at lane `first_behind` (> 0): the lanes from there on count from its anchor
					const unsigned first_behind = (unsigned) ((next_boundary - bounds[k].start) / traits::E);
					const scan_type at_boundary = wave64::read_lane(rows[k].prefix[traits::E - 1], first_behind - 1);
					const U behind = (U) (chunk_anchor_values.of(segment++) - (U) at_boundary);
					rows[k].store(out, bounds[k].start, bounds[k].end, lane, lane < first_behind ? base : behind, patches);
					base = behind;
					next_boundary += L;
				}
				else { rows[k].store(out, bounds[k].start, bounds[k].end, lane, base, patches); }
				base += (U) rows[k].total;
			}
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { rows[k].deltas = next[k].deltas; bounds[k] = next_bounds[k]; }
		}
	}
}

/// Segments SHORTER than a row (round 4): several of them in every row.  wave_per_ragged_segment gives each its own row — segments of 64 elements
/// of 4 bytes fill a quarter of the lanes and of every store (20 % of the HBM peak; 16-element segments: 6 %).  Here, as above, rows follow the
/// output's cache lines whatever the segments do, and every LANE works out where it stands: its group of E elements lies in ONE segment (segment
/// lengths are multiples of E), segment `s` of the chunk; if that segment begins in this row, at lane b, the lane's values count from the
/// segment's anchor less the row's prefix before lane b (one ds_bpermute); if it began in an earlier row they go on from where that row ended.
/// The anchors are gathered per lane with the row's own loads (lanes of one segment read one address).  A chunk is as many whole segments
/// as fill `rows_per_chunk` rows (short_segments_per_wave_chunk; kernel and launch parameters agree).
template <unsigned UncompressedSize, unsigned CompressedSize>
__host__ __device__ inline bool short_segments_share_rows(uint64_t L)
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	return L % traits::E == 0 && L % traits::row != 0 && (L < traits::row || (L + traits::row - 1) / traits::row <= rows_per_chunk);
}
/// segments per wavefront chunk, of any length (the wave path)
template <unsigned UncompressedSize, unsigned CompressedSize>
__host__ __device__ inline uint64_t segments_per_wave_chunk_of(uint64_t L)
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	if (L < traits::row && short_segments_share_rows<UncompressedSize, CompressedSize>(L)) { return (uint64_t) rows_per_chunk * traits::row / L; }
	return segments_per_wave_chunk((L + traits::row - 1) / traits::row);
}

/// Chunks of the lane-placed path.  A wavefront's chunk had better be EXACTLY rows_per_chunk rows (measured in round 4, docs/history/r04.md, with
/// builds with other values of rows_per_chunk: chunks of 16 or 32 rows 73 % of the HBM peak, of 21, 24, 30, 31, 33, 48 rows 62-66 % — whole-row segments
/// of 10 or 12 rows, three or two to a chunk, sit there too; neither the base addresses' alignment nor the number of workgroups explains it,
/// and the bytes moved are the same).  So where whole segments do not make such a chunk the chunks are WINDOWS of rows_per_chunk rows of the column
/// whatever the segments do: a window that begins inside a segment first adds up that segment's deltas before the window (at most one segment's
/// worth of input read a second time — segments of up to windows_up_to_rows rows go this way).
enum : unsigned { windows_up_to_rows = 4 };   // (measured with 16: segments of 8-12 rows lose 3-4 points to the second read, of up to 4 rows gain 5-13)
template <unsigned UncompressedSize, unsigned CompressedSize>
__host__ __device__ inline bool chunks_are_windows(uint64_t L)
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	return short_segments_share_rows<UncompressedSize, CompressedSize>(L) && L <= (uint64_t) windows_up_to_rows * traits::row
		&& segments_per_wave_chunk_of<UncompressedSize, CompressedSize>(L) * L != (uint64_t) rows_per_chunk * traits::row;
}
/// elements per wavefront chunk on the wave path, for segments of any length
template <unsigned UncompressedSize, unsigned CompressedSize>
__host__ __device__ inline uint64_t wave_chunk_length_of(uint64_t L)
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	if (chunks_are_windows<UncompressedSize, CompressedSize>(L)) { return (uint64_t) rows_per_chunk * traits::row; }
	return segments_per_wave_chunk_of<UncompressedSize, CompressedSize>(L) * L;
}

/// `Staged`: the chunk's input — exactly the staging area's rows_per_chunk rows, beginning 16-byte aligned: the caller checks — goes through wave-private
/// LDS first, in 16-byte loads (64-bit elements from 1-2-byte deltas, whose rows are 2-4 bytes per lane); the rows' ANCHORS are still gathered
/// a few rows ahead of their use.
template <unsigned UncompressedSize, unsigned CompressedSize, bool Staged = false, typename Patches, typename Grid = launch_grid::whole>
__device__ __forceinline__ void wave_rows_of_short_segments(
	uint_t<UncompressedSize>* __restrict__ out, const uint_t<CompressedSize>* __restrict__ in,
	const uint_t<UncompressedSize>* __restrict__ anchors, uint64_t n, uint64_t L, Patches patches, uint8_t* staging = nullptr,
	Grid grid = Grid())
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	using U = typename traits::uncompressed_type;
	using C = typename traits::compressed_type;
	using scan_type = typename traits::scan_type;
	typedef uint32_t piece_t __attribute__((ext_vector_type(4)));
	constexpr unsigned R = Staged ? (Patches::enabled ? 2 : 4) : (traits::rows_in_flight > 4 ? 4 : traits::rows_in_flight);   // (8 rows of 2-byte elements with their bounds in flight spill registers)
	constexpr unsigned log2_E = traits::E == 8 ? 3 : (traits::E == 4 ? 2 : 1);
	static_assert((1u << log2_E) == traits::E, "E is 2, 4 or 8");
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) grid.block() * wave64::per_block() + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) grid.blocks() * wave64::per_block();
	const uint64_t segments_per_chunk = segments_per_wave_chunk_of<UncompressedSize, CompressedSize>(L);
	const bool windows = chunks_are_windows<UncompressedSize, CompressedSize>(L);
	const uint64_t chunk_length = wave_chunk_length_of<UncompressedSize, CompressedSize>(L);   // at most rows_per_chunk rows: 32-bit positions within a chunk
	const uint64_t num_chunks = (n + chunk_length - 1) / chunk_length;
	const unsigned segment_length = (unsigned) L;
	const float segments_per_element = 1.0f / (float) segment_length;
	using sum_type = typename std::conditional<(sizeof(U) < 4), uint32_t, U>::type;

	struct row_bounds { uint64_t start, end; };   // wave-uniform
	struct row_places { unsigned segment; U anchor; };   // per lane: the segment (of the chunk) its elements are in, and that segment's anchor
	for (uint64_t c = wave; c < num_chunks; c += num_waves) {
		const uint64_t chunk_start = c * chunk_length;
		const uint64_t chunk_end = (chunk_start + chunk_length < n) ? chunk_start + chunk_length : n;
		const uint64_t first_segment = windows ? chunk_start / L : c * segments_per_chunk;
		const uint64_t origin = first_segment * L;   // where the chunk's first segment begins: at the chunk's start, or (windows) before it
		const uint64_t misplaced = (reinterpret_cast<uintptr_t>(out + chunk_start) / sizeof(U)) % traits::row;
		const uint64_t head = misplaced == 0 ? traits::row : traits::row - misplaced;
		uint64_t walk_start = chunk_start;
		auto take_row = [&]() {
			row_bounds r;
			r.start = walk_start;
			const uint64_t length = walk_start == chunk_start ? head : traits::row;
			r.end = walk_start + length < chunk_end ? walk_start + length : chunk_end;
			walk_start = r.end;
			return r;   // past the chunk's end: start == end == chunk_end, an empty row
		};
		auto place = [&](const row_bounds& b) {
			row_places where;
			const uint64_t p = b.start + (uint64_t) lane * traits::E;
			const unsigned past_origin = (unsigned) (p - origin);
			// (past_origin < 2^16, segment_length >= 2: the quotient is exact in single precision — the +0.5 keeps it 0.5 / L away from the integers)
			where.segment = (unsigned) (((float) past_origin + 0.5f) * segments_per_element);
			where.anchor = p < b.end ? anchors[first_segment + where.segment] : (U) 0;
			return where;
		};
		U carry = 0;   // the value before this row's first element (of a segment that began in an earlier row)
		row_t<UncompressedSize, CompressedSize> rows[R], next[R];
		row_bounds bounds[R], next_bounds[R];
		row_places places[R], next_places[R];
		patches.begin_seek(chunk_start, lane);
		#pragma unroll
		for (unsigned k = 0; k < R; k++) { bounds[k] = take_row(); if (!Staged) { rows[k].load(in, bounds[k].start, bounds[k].end, lane); } places[k] = place(bounds[k]); }
		if (Staged) {
			const unsigned chunk_bytes = (unsigned) (chunk_end - chunk_start) * CompressedSize;
			const uint8_t* from = reinterpret_cast<const uint8_t*>(in) + chunk_start * CompressedSize;
			wave64::lds_fence();   // (the previous chunk's rows have been read)
			for (unsigned at = lane * 16; at < chunk_bytes; at += 64 * 16) {
				if (at + 16 <= chunk_bytes) { *reinterpret_cast<piece_t*>(staging + at) = *reinterpret_cast<const piece_t*>(from + at); }
				else { for (unsigned b = at; b < chunk_bytes; b++) { staging[b] = from[b]; } }
			}
		}
		if (origin < chunk_start) {   // (wave-uniform) a window that begins inside a segment: the value before its first element
			sum_type sum = 0;
			for (uint64_t at = origin; at < chunk_start; at += 4 * traits::row) {
				row_t<UncompressedSize, CompressedSize> before[4];
				#pragma unroll
				for (unsigned k = 0; k < 4; k++) { before[k].load(in, at + (uint64_t) k * traits::row, chunk_start, lane); }
				#pragma unroll
				for (unsigned k = 0; k < 4; k++) {
					#pragma unroll
					for (unsigned j = 0; j < traits::E; j++) { sum += (sum_type) before[k].deltas[j]; }
				}
			}
			carry = (U) (anchors[first_segment] + (U) wave64::read_last_lane(wave64::inclusive_sum(sum)));
		}
		patches.finish_seek(lane);
		if (Staged) { wave64::lds_fence(); }
		while (bounds[0].start < chunk_end) {
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { next_bounds[k] = take_row(); if (!Staged) { next[k].load(in, next_bounds[k].start, next_bounds[k].end, lane); } next_places[k] = place(next_bounds[k]); }
			if (Staged) {
				#pragma unroll
				for (unsigned k = 0; k < R; k++) {
					const uint64_t p = bounds[k].start + (uint64_t) lane * traits::E;
					const C* staged = reinterpret_cast<const C*>(staging + (p - chunk_start) * CompressedSize);
					#pragma unroll
					for (unsigned j = 0; j < traits::E; j++) { rows[k].deltas[j] = (p + j < bounds[k].end) ? staged[j] : (C) 0; }
				}
			}
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { rows[k].scan(); }
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				if (bounds[k].start >= chunk_end) { continue; }   // wave-uniform
				const unsigned row_within_chunk = (unsigned) (bounds[k].start - origin);   // (both from the first segment's start)
				const unsigned segment_within_chunk = places[k].segment * segment_length;   // where the lane's segment begins
				const bool begins_here = segment_within_chunk >= row_within_chunk;
				const unsigned first_lane = begins_here ? (segment_within_chunk - row_within_chunk) >> log2_E : 0u;
				const scan_type up_to_here = rows[k].prefix[traits::E - 1];   // the row's prefix up to the lane's last element
				const scan_type before_segment = wave64::read_lane_of_choice(up_to_here, (first_lane + 63u) & 63u);   // (lane 0: not used)
				const U base = begins_here ? (U) (places[k].anchor - (first_lane == 0 ? (U) 0 : (U) before_segment)) : carry;
				rows[k].store(out, bounds[k].start, bounds[k].end, lane, base, patches);
				const unsigned last_lane = (unsigned) ((bounds[k].end - bounds[k].start) >> log2_E) - 1;
				carry = wave64::read_lane((U) (base + (U) up_to_here), last_lane);
			}
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { if (!Staged) { rows[k].deltas = next[k].deltas; } bounds[k] = next_bounds[k]; places[k] = next_places[k]; }
		}
	}
}

/// ... and with the chunk's input staged through wave-private LDS first (as wave_per_segment_with_staged_input does for whole-row segments): for the
/// instantiations whose rows are only 2-4 input bytes per lane — 64-bit values from 1-2-byte deltas — element-sized loads per row are what limits
/// the ragged paths (59-61 % with rows by segment or by cache line alike).  `in` must be 16-byte aligned and the chunk leave 16 bytes of the staging free.
template <unsigned UncompressedSize, unsigned CompressedSize, typename Patches, typename Grid = launch_grid::whole>
__device__ __forceinline__ void wave_per_ragged_segment_with_staged_input(
	uint_t<UncompressedSize>* __restrict__ out, const uint_t<CompressedSize>* __restrict__ in,
	const uint_t<UncompressedSize>* __restrict__ anchors, uint64_t n, uint64_t L, Patches patches, uint8_t* staging,
	Grid grid = Grid())
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	using U = typename traits::uncompressed_type;
	using C = typename traits::compressed_type;
	using scan_type = typename traits::scan_type;
	typedef uint32_t piece_t __attribute__((ext_vector_type(4)));
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) grid.block() * wave64::per_block() + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) grid.blocks() * wave64::per_block();
	const uint64_t segments_per_chunk = segments_per_wave_chunk((L + traits::row - 1) / traits::row);
	const uint64_t chunk_length = segments_per_chunk * L;
	const uint64_t num_chunks = (n + chunk_length - 1) / chunk_length;
	const uint8_t* in_bytes = reinterpret_cast<const uint8_t*>(in);
	constexpr uint64_t piece_capacity = (rows_per_chunk * traits::row * CompressedSize - 16) / CompressedSize / traits::E * traits::E;   // elements (16 bytes are the lead's)

	for (uint64_t c = wave; c < num_chunks; c += num_waves) {
		const uint64_t start = c * chunk_length;
		const uint64_t end = (start + chunk_length < n) ? start + chunk_length : n;
		uint64_t segment = c * segments_per_chunk, next_boundary = start;   // the next segment's first element
		U base = 0;
		chunk_anchors<U> chunk_anchor_values;
		chunk_anchor_values.load(anchors, segment, segments_per_chunk, start, end, L, lane);
		patches.begin_seek(start, lane);
		// the staging holds a PIECE of the chunk at a time — all of it when the chunk is of up to rows_per_chunk rows; a longer segment's later
		// pieces begin on row-sized boundaries of the output
		uint64_t piece_end = 0;
		for (uint64_t piece_start = start; piece_start < end; piece_start = piece_end) {
			piece_end = piece_start + piece_capacity;
			if (piece_end < end) { piece_end -= (reinterpret_cast<uintptr_t>(out + piece_end) / sizeof(U)) % traits::row; } else { piece_end = end; }
			// (a piece's bytes need not begin on a 16-byte boundary: the `lead` bytes before them come along — the column's own, `in` is aligned)
			const unsigned lead = (unsigned) ((piece_start * CompressedSize) % 16);
			const unsigned piece_bytes = lead + (unsigned) (piece_end - piece_start) * CompressedSize;
			const uint8_t* from = in_bytes + piece_start * CompressedSize - lead;
			wave64::lds_fence();   // (the previous piece's rows have been read)
			for (unsigned at = lane * 16; at < piece_bytes; at += 64 * 16) {
				if (at + 16 <= piece_bytes) {
					*reinterpret_cast<piece_t*>(staging + at) = *reinterpret_cast<const piece_t*>(from + at);
				}
				else {
					for (unsigned b = at; b < piece_bytes; b++) { staging[b] = from[b]; }
				}
			}
			if (piece_start == start) { patches.finish_seek(lane); }
			wave64::lds_fence();
			const uint64_t misplaced = (reinterpret_cast<uintptr_t>(out + piece_start) / sizeof(U)) % traits::row;
			uint64_t row_end = 0;
			for (uint64_t row_start = piece_start; row_start < piece_end; row_start = row_end) {
				const uint64_t length = row_start == piece_start && misplaced != 0 ? traits::row - misplaced : traits::row;   // rows begin on row-sized boundaries of the output
				row_end = row_start + length < piece_end ? row_start + length : piece_end;
				row_t<UncompressedSize, CompressedSize> row;
				const uint64_t p = row_start + (uint64_t) lane * traits::E;
				const C* staged = reinterpret_cast<const C*>(staging + lead + (p - piece_start) * CompressedSize);
				#pragma unroll
				for (unsigned j = 0; j < traits::E; j++) { row.deltas[j] = (p + j < row_end) ? staged[j] : (C) 0; }
				row.scan();
				if (row_start == next_boundary) { base = chunk_anchor_values.of(segment++); next_boundary += L; }   // wave-uniform
				if (next_boundary < row_end) {
					const unsigned first_behind = (unsigned) ((next_boundary - row_start) / traits::E);
					const scan_type at_boundary = wave64::read_lane(row.prefix[traits::E - 1], first_behind - 1);
					const U behind = (U) (chunk_anchor_values.of(segment++) - (U) at_boundary);
					row.store(out, row_start, row_end, lane, lane < first_behind ? base : behind, patches);
					base = behind;
					next_boundary += L;
				}
				else { row.store(out, row_start, row_end, lane, base, patches); }
				base += (U) row.total;
			}
		}
	}
}

/// Segment lengths that are NOT multiples of a lane's E elements (round 4) — 6, 10, 1001, 4094 ...: a lane's group straddles segment boundaries, the 16-byte
/// paths above do not apply, and the element-wise `segmented` path below runs at 13-17 % of the HBM peak.  Here, for segments of up to rows_per_chunk
/// rows: chunks are WINDOWS of rows_per_chunk rows of the column (not of segments), rows follow the output's 1 KiB boundaries, loads and stores are 16
/// bytes per lane as everywhere — and every ELEMENT finds its segment: an element that begins one (position % L == 0; its lane gathers the anchor with
/// the row's loads) leaves `anchor - the row's prefix before it` in a wave-private LDS table indexed by the segment's number within the row; every element
/// then reads its segment's entry — or, if its segment began before the row, goes on from where the previous row ended.  A window that begins inside a
/// segment first adds up that segment's deltas before it (the input from the segment's start to the window: read twice).
enum : unsigned { any_length_waves = 4 };   // wavefronts per workgroup the LDS tables exist for
template <unsigned UncompressedSize, unsigned CompressedSize>
__host__ __device__ inline bool any_length_by_windows(uint64_t L)
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	return L % traits::E != 0 && L <= (uint64_t) rows_per_chunk * traits::row;
}

/// `AnyAlignment` (decompress_at_any_alignment): `out` and `in` aligned for their elements only.  Rows follow the output's 1 KiB boundaries as before — every row but the
/// window's first is then stored 16 bytes per lane at aligned addresses, wherever the column begins; the first one, up to the boundary, goes element by element; the
/// rows' loads are 16 bytes per lane at whatever address (gfx950 takes them).
template <unsigned UncompressedSize, unsigned CompressedSize, typename Patches, bool AnyAlignment = false, typename Grid = launch_grid::whole>
__device__ __forceinline__ void wave_windows_of_any_segments(
	uint_t<UncompressedSize>* __restrict__ out, const uint_t<CompressedSize>* __restrict__ in,
	const uint_t<UncompressedSize>* __restrict__ anchors, uint64_t n, uint64_t L, Patches patches, uint_t<UncompressedSize>* segment_bases,   // (LDS: traits::row entries, this wavefront's)
	Grid grid = Grid())
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	using U = typename traits::uncompressed_type;
	using C = typename traits::compressed_type;
	using scan_type = typename traits::scan_type;
	using sum_type = typename std::conditional<(sizeof(U) < 4), uint32_t, U>::type;
	constexpr unsigned E = traits::E;
	constexpr unsigned R = Patches::enabled ? 1 : 2;   // (this path's registers count for every path of the kernel: the fused kernels' cursors leave room for one row in flight)
	constexpr unsigned log2_E = E == 8 ? 3 : (E == 4 ? 2 : 1);
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) grid.block() * wave64::per_block() + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) grid.blocks() * wave64::per_block();
	constexpr uint64_t window = (uint64_t) rows_per_chunk * traits::row;
	const uint64_t num_chunks = (n + window - 1) / window;
	const unsigned segment_length = (unsigned) L;
	const float segments_per_element = 1.0f / (float) segment_length;

	struct row_bounds { uint64_t start, end; };   // wave-uniform
	struct row_places { unsigned segment, within; U start_anchor[E]; };   // per lane: its first element's segment (of the chunk) and place in it; the anchors of the segments its elements begin
	for (uint64_t c = wave; c < num_chunks; c += num_waves) {
		const uint64_t chunk_start = c * window;
		const uint64_t chunk_end = chunk_start + window < n ? chunk_start + window : n;
		const uint64_t first_segment = ((chunk_start | L) >> 32) == 0 ? (uint64_t) ((uint32_t) chunk_start / (uint32_t) L) : chunk_start / L;
		const uint64_t origin = first_segment * L;   // (at most a segment — rows_per_chunk rows — before the window: positions from it fit 16 bits)
		const uint64_t misplaced = (reinterpret_cast<uintptr_t>(out + chunk_start) / sizeof(U)) % traits::row;
		const uint64_t head = misplaced == 0 ? traits::row : traits::row - misplaced;
		uint64_t walk_start = chunk_start;
		auto take_row = [&]() {
			row_bounds r;
			r.start = walk_start;
			const uint64_t length = walk_start == chunk_start ? head : traits::row;
			r.end = walk_start + length < chunk_end ? walk_start + length : chunk_end;
			walk_start = r.end;
			return r;
		};
		auto place = [&](const row_bounds& b) {
			row_places where;
			const uint64_t p = b.start + (uint64_t) lane * E;
			const unsigned past_origin = (unsigned) (p - origin);
			where.segment = (unsigned) (((float) past_origin + 0.5f) * segments_per_element);   // (exact: see wave_rows_of_short_segments)
			where.within = past_origin - where.segment * segment_length;
			unsigned segment = where.segment, within = where.within;
			#pragma unroll
			for (unsigned j = 0; j < E; j++) {
				where.start_anchor[j] = (p + j < b.end && within == 0) ? anchors[first_segment + segment] : (U) 0;
				if (++within == segment_length) { within = 0; segment++; }
			}
			return where;
		};
		U carry = 0;
		row_t<UncompressedSize, CompressedSize> rows[R], next[R];
		row_bounds bounds[R], next_bounds[R];
		row_places places[R], next_places[R];
		patches.begin_seek(chunk_start, lane);
		#pragma unroll
		for (unsigned k = 0; k < R; k++) { bounds[k] = take_row(); rows[k].template load<AnyAlignment>(in, bounds[k].start, bounds[k].end, lane); places[k] = place(bounds[k]); }
		if (origin < chunk_start) {   // (wave-uniform) the window begins inside a segment: the value before its first element
			sum_type sum = 0;
			const uint64_t aligned_origin = origin / E * E;   // (16-byte loads: the elements of the first group that lie before the segment are left out)
			for (uint64_t at = aligned_origin; at < chunk_start; at += 4 * traits::row) {
				row_t<UncompressedSize, CompressedSize> before[4];
				#pragma unroll
				for (unsigned k = 0; k < 4; k++) { before[k].template load<AnyAlignment>(in, at + (uint64_t) k * traits::row, chunk_start, lane); }
				#pragma unroll
				for (unsigned k = 0; k < 4; k++) {
					const uint64_t q = at + (uint64_t) k * traits::row + (uint64_t) lane * E;
					#pragma unroll
					for (unsigned j = 0; j < E; j++) { sum += (q + j >= origin) ? (sum_type) before[k].deltas[j] : (sum_type) 0; }
				}
			}
			carry = (U) (anchors[first_segment] + (U) wave64::read_last_lane(wave64::inclusive_sum(sum)));
		}
		patches.finish_seek(lane);
		while (bounds[0].start < chunk_end) {
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { next_bounds[k] = take_row(); next[k].template load<AnyAlignment>(in, next_bounds[k].start, next_bounds[k].end, lane); next_places[k] = place(next_bounds[k]); }
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { rows[k].scan(); }
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				if (bounds[k].start >= chunk_end) { continue; }   // wave-uniform
				const uint64_t p = bounds[k].start + (uint64_t) lane * E;
				const unsigned row_past_origin = (unsigned) (bounds[k].start - origin);
				const unsigned first_segment_of_row = wave64::read_first_lane(places[k].segment);
				// 1. the elements that begin a segment leave its base — anchor less the row's prefix before it — in the table
				unsigned segment = places[k].segment, within = places[k].within;
				unsigned segment_of[E]; bool begins_in_row[E];
				#pragma unroll
				for (unsigned j = 0; j < E; j++) {
					const scan_type before_element = rows[k].prefix[j] - (scan_type) rows[k].deltas[j];
					if (p + j < bounds[k].end && within == 0) { segment_bases[segment - first_segment_of_row] = (U) (places[k].start_anchor[j] - (U) before_element); }
					segment_of[j] = segment;
					begins_in_row[j] = (unsigned) (p - origin) + j - within >= row_past_origin;   // where the element's segment begins, against where the row does
					if (++within == segment_length) { within = 0; segment++; }
				}
				wave64::lds_fence();
				// 2. every element: its segment's base, or on from the previous row
				memory_access::group<U, E> values;
				#pragma unroll
				for (unsigned j = 0; j < E; j++) {
					const U base = begins_in_row[j] ? segment_bases[segment_of[j] - first_segment_of_row] : carry;
					values[j] = (U) (base + (U) rows[k].prefix[j]);
				}
				wave64::lds_fence();   // (the table is rewritten for the next row)
				const unsigned last_lane = (unsigned) ((bounds[k].end - bounds[k].start + E - 1) >> log2_E) - 1;   // (a row is whole groups but at the column's end)
				const unsigned last_slot = (unsigned) ((bounds[k].end - bounds[k].start - 1) & (E - 1));
				U last_value = values[0];
				#pragma unroll
				for (unsigned j = 1; j < E; j++) { if (last_slot == j) { last_value = values[j]; } }
				carry = wave64::read_lane(last_value, last_lane);
				if (AnyAlignment && bounds[k].start == chunk_start && !memory_access::is_aligned(out + chunk_start, 16)) {   // (wave-uniform: the row up to the output's boundary)
					patches.template apply_to_row<E>(bounds[k].start, bounds[k].end, lane, values);
					#pragma unroll
					for (unsigned j = 0; j < E; j++) { if (p + j < bounds[k].end) { out[p + j] = values[j]; } }
				}
				else { rows[k].store_values(out, bounds[k].start, bounds[k].end, lane, values, patches); }
			}
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { rows[k].deltas = next[k].deltas; bounds[k] = next_bounds[k]; places[k] = next_places[k]; }
		}
	}
}

/// Segments LONGER than a window, at any alignment (decompress_windows_at_any_alignment): windows of rows_per_chunk rows that follow the output's 1 KiB boundaries as above,
/// the value before each window's first element given (`window_bases`, from the pre-pass) — and, since a window holds at most ONE segment start, everything in registers:
/// an element at or behind the start counts from the new segment's anchor less the row's prefix before the start (one read_lane), the others go on from the carry.
template <unsigned UncompressedSize, unsigned CompressedSize>
__device__ __forceinline__ void wave_windows_of_long_segments(
	uint_t<UncompressedSize>* __restrict__ out, const uint_t<CompressedSize>* __restrict__ in,
	const uint_t<UncompressedSize>* __restrict__ anchors, uint64_t n, uint64_t L, const uint_t<UncompressedSize>* __restrict__ window_bases)
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	using U = typename traits::uncompressed_type;
	using scan_type = typename traits::scan_type;
	constexpr unsigned E = traits::E;
	constexpr unsigned R = 2;
	constexpr unsigned log2_E = E == 8 ? 3 : (E == 4 ? 2 : 1);
	constexpr uint64_t window = (uint64_t) rows_per_chunk * traits::row;
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) blockIdx.x * wave64::per_block() + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) gridDim.x * wave64::per_block();
	const uint64_t num_windows = (n + window - 1) / window;
	sorted_patches::none no_patches;
	struct row_bounds { uint64_t start, end; };   // wave-uniform

	for (uint64_t c = wave; c < num_windows; c += num_waves) {
		const uint64_t chunk_start = c * window;
		const uint64_t chunk_end = chunk_start + window < n ? chunk_start + window : n;
		const uint64_t first_segment = chunk_start / L, origin = first_segment * L;
		// the one segment start the window may hold: its own first element, or `origin + L` inside it
		const bool begins_a_segment = origin == chunk_start;
		const uint64_t boundary = begins_a_segment ? chunk_start : origin + L;   // (>= chunk_end: none)
		const bool has_boundary = boundary < chunk_end;
		U anchor_at_boundary = 0;
		if (has_boundary) { anchor_at_boundary = anchors[begins_a_segment ? first_segment : first_segment + 1]; }   // wave-uniform: a scalar load
		U carry = 0;
		if (!begins_a_segment) { carry = window_bases[c]; }
		const uint64_t misplaced = (reinterpret_cast<uintptr_t>(out + chunk_start) / sizeof(U)) % traits::row;
		const uint64_t head = misplaced == 0 ? traits::row : traits::row - misplaced;
		uint64_t walk_start = chunk_start;
		auto take_row = [&]() {
			row_bounds r;
			r.start = walk_start;
			const uint64_t length = walk_start == chunk_start ? head : traits::row;
			r.end = walk_start + length < chunk_end ? walk_start + length : chunk_end;
			walk_start = r.end;
			return r;
		};
		row_t<UncompressedSize, CompressedSize> rows[R], next[R];
		row_bounds bounds[R], next_bounds[R];
		#pragma unroll
		for (unsigned k = 0; k < R; k++) { bounds[k] = take_row(); rows[k].template load<true>(in, bounds[k].start, bounds[k].end, lane); }
		while (bounds[0].start < chunk_end) {
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { next_bounds[k] = take_row(); next[k].template load<true>(in, next_bounds[k].start, next_bounds[k].end, lane); }
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { rows[k].scan(); }
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				if (bounds[k].start >= chunk_end) { continue; }   // wave-uniform
				const uint64_t p = bounds[k].start + (uint64_t) lane * E;
				const bool boundary_in_row = has_boundary && boundary >= bounds[k].start && boundary < bounds[k].end;   // wave-uniform
				memory_access::group<U, E> values;
				U new_base = 0;
				unsigned boundary_in_lanes = 0xFFFFFFFFu;   // the boundary's place in the row, in elements
				if (boundary_in_row) {
					boundary_in_lanes = (unsigned) (boundary - bounds[k].start);
					// the row's prefix before the boundary: the inclusive prefix of the element before it (0 at the row's first element)
					scan_type before = 0;
					if (boundary_in_lanes != 0) {
						const unsigned at = boundary_in_lanes - 1, slot = at & (E - 1);
						scan_type mine = rows[k].prefix[0];
						#pragma unroll
						for (unsigned j = 1; j < E; j++) { if (slot == j) { mine = rows[k].prefix[j]; } }
						before = wave64::read_lane(mine, at >> log2_E);
					}
					new_base = (U) (anchor_at_boundary - (U) before);
				}
				const unsigned q = lane * E;
				#pragma unroll
				for (unsigned j = 0; j < E; j++) { values[j] = (U) ((q + j >= boundary_in_lanes ? new_base : carry) + (U) rows[k].prefix[j]); }
				// the value of the row's last element: the next row's carry
				const unsigned last_lane = (unsigned) ((bounds[k].end - bounds[k].start + E - 1) >> log2_E) - 1;
				const unsigned last_slot = (unsigned) ((bounds[k].end - bounds[k].start - 1) & (E - 1));
				U last_value = values[0];
				#pragma unroll
				for (unsigned j = 1; j < E; j++) { if (last_slot == j) { last_value = values[j]; } }
				carry = wave64::read_lane(last_value, last_lane);
				if (bounds[k].start == chunk_start && !memory_access::is_aligned(out + chunk_start, 16)) {   // (wave-uniform: the row up to the output's boundary)
					#pragma unroll
					for (unsigned j = 0; j < E; j++) { if (p + j < bounds[k].end) { out[p + j] = values[j]; } }
				}
				else { rows[k].store_values(out, bounds[k].start, bounds[k].end, lane, values, no_patches); }
			}
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { rows[k].deltas = next[k].deltas; bounds[k] = next_bounds[k]; }
		}
	}
}

/// block path: the workgroup's wavefronts sweep one segment together — wavefront w takes R consecutive
/// rows of every sweep, so a segment of up to waves*R rows needs ONE barrier.  (Round 4 measured every wavefront taking every num_waves-th row instead — the workgroup
/// reading and writing num_waves consecutive rows at every step, as the fixed-rate decoders' tiles do: slower or equal everywhere (profiles/r04_delta_segment_lengths.txt),
/// and its per-row totals cost every path of the kernel 25 registers; not kept.)
template <unsigned UncompressedSize, unsigned CompressedSize, typename Patches, typename Grid = launch_grid::whole>
__device__ __forceinline__ void block_per_segment(
	uint_t<UncompressedSize>* __restrict__ out, const uint_t<CompressedSize>* __restrict__ in,
	const uint_t<UncompressedSize>* __restrict__ anchors, uint64_t n, uint64_t L, Patches patches,
	Grid grid = Grid())
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	using U = typename traits::uncompressed_type;
	constexpr unsigned R = rows_per_wave_per_sweep;
	__shared__ U wave_totals[2][max_waves_per_block];
	const unsigned lane = wave64::lane_id();
	const unsigned w = wave64::index_in_block();
	const unsigned num_waves = wave64::per_block();
	const uint64_t num_segments = (n + L - 1) / L;
	unsigned buffer = 0;

	// Segments of at most half the workgroup's rows (whole-row segments of 8 or 16 rows under a 4-wavefront workgroup): SEVERAL segments per workgroup, each
	// taken by 1 or 2 of its wavefronts in one sweep — the same barrier, the totals added up within a segment's wavefronts only
	const uint64_t rows_per_segment = (L + traits::row - 1) / traits::row;
	const unsigned waves_per_segment = (unsigned) ((rows_per_segment + R - 1) / R);
	if (waves_per_segment * 2 <= num_waves) {
		const unsigned segments_per_block = num_waves / waves_per_segment;
		const unsigned mine_of_block = w / waves_per_segment, part = w % waves_per_segment;
		for (uint64_t first = (uint64_t) grid.block() * segments_per_block; first < num_segments; first += (uint64_t) grid.blocks() * segments_per_block) {
			const uint64_t s = first + mine_of_block;
			const bool working = mine_of_block < segments_per_block && s < num_segments;
			const uint64_t start = s * L, end = working ? ((start + L < n) ? start + L : n) : start;
			const uint64_t mine = start + (uint64_t) part * R * traits::row;
			row_t<UncompressedSize, CompressedSize> rows[R];
			const U base = working ? anchors[s] : (U) 0;
			if (working && mine < end) { patches.begin_seek(mine, lane); }
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { rows[k].load(in, mine + (uint64_t) k * traits::row, end, lane); }   // (idle wavefronts, rows past `end`: zeros)
			if (working && mine < end) { patches.finish_seek(lane); }
			U my_total = 0;
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { rows[k].scan(); my_total += (U) rows[k].total; }
			if (lane == 0) { wave_totals[buffer][w] = my_total; }
			__syncthreads();
			U running = base;
			for (unsigned v = w - part; v < w; v++) { running += wave_totals[buffer][v]; }
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				rows[k].store(out, mine + (uint64_t) k * traits::row, end, lane, running, patches);
				running += (U) rows[k].total;
			}
			buffer ^= 1;
		}
		return;
	}

	for (uint64_t s = grid.block(); s < num_segments; s += grid.blocks()) {
		const uint64_t start = s * L;
		const uint64_t end = (start + L < n) ? start + L : n;
		U base = anchors[s];
		for (uint64_t sweep_start = start; sweep_start < end; sweep_start += (uint64_t) num_waves * R * traits::row) {
			const uint64_t mine = sweep_start + (uint64_t) w * R * traits::row;
			row_t<UncompressedSize, CompressedSize> rows[R];
			// a wavefront seeks to ITS rows of the sweep, the probes in flight with the rows' own loads (walking there from the
			// segment's start costs a serial step per patch of every wavefront below, every sweep)
			if (mine < end) { patches.begin_seek(mine, lane); }
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { rows[k].load(in, mine + (uint64_t) k * traits::row, end, lane); }   // rows past `end` load zeros
			if (mine < end) { patches.finish_seek(lane); }
			U my_total = 0;
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { rows[k].scan(); my_total += (U) rows[k].total; }
			if (lane == 0) { wave_totals[buffer][w] = my_total; }
			__syncthreads();
			U before = 0, all = 0;
			for (unsigned v = 0; v < num_waves; v++) {
				U t = wave_totals[buffer][v];
				if (v < w) { before += t; }
				all += t;
			}
			U running = (U) (base + before);
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				rows[k].store(out, mine + (uint64_t) k * traits::row, end, lane, running, patches);
				running += (U) rows[k].total;
			}
			base += all;
			buffer ^= 1;   // the next sweep writes the other table; this one is re-written only after another barrier
		}
	}
}

/// general path: any segment length, any alignment — segmented scan with reset at segment heads
template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize, typename Patches, typename Grid = launch_grid::whole>
__device__ __forceinline__ void segmented(
	uint_t<UncompressedSize>* __restrict__ out, const uint_t<CompressedSize>* __restrict__ in,
	const uint_t<UncompressedSize>* __restrict__ anchors, uint64_t n, uint64_t L, Patches patches,
	Grid grid = Grid())
{
	using U = uint_t<UncompressedSize>;
	enum : unsigned { E = 4, row = 64 * E };
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) grid.block() * wave64::per_block() + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) grid.blocks() * wave64::per_block();
	// a wavefront owns a run of whole segments spanning at least one row
	const uint64_t segments_per_chunk = L >= row ? 1 : (row + L - 1) / L;
	const uint64_t chunk_length = segments_per_chunk * L;
	const uint64_t num_chunks = (n + chunk_length - 1) / chunk_length;

	for (uint64_t c = wave; c < num_chunks; c += num_waves) {
		const uint64_t start = c * chunk_length;
		const uint64_t end = (start + chunk_length < n) ? start + chunk_length : n;
		U carry = 0;   // sum since the last segment head before the current row (the chunk starts on a head)
		patches.seek(start, lane);
		for (uint64_t row_start = start; row_start < end; row_start += row) {
			const uint64_t p0 = row_start + (uint64_t) lane * E;
			uint64_t seg, within;
			if (IndexSize <= 4 || ((p0 | L) >> 32) == 0) { uint32_t q = (uint32_t) p0 / (uint32_t) L; seg = q; within = (uint32_t) p0 - q * (uint32_t) L; }
			else { seg = p0 / L; within = p0 - seg * L; }
			U value[E]; uint64_t seg_of[E]; bool before_head[E];
			U running = 0; bool has_head = false;
			#pragma unroll
			for (unsigned j = 0; j < E; j++) {
				U d = (p0 + j < end) ? (U) in[p0 + j] : (U) 0;
				if (within == 0) { running = 0; has_head = true; }
				running += d;
				value[j] = running; seg_of[j] = seg; before_head[j] = !has_head;
				if (++within == L) { within = 0; seg++; }
			}
			// wave-wide segmented inclusive scan of (has_head, running): Kogge-Stone over shuffles
			U acc = running; bool flag = has_head;
			#pragma unroll
			for (unsigned d = 1; d < 64; d <<= 1) {
				U other = __shfl_up(acc, d, 64);
				int other_flag = __shfl_up((int) flag, d, 64);
				if (lane >= d) { if (!flag) { acc += other; } flag = flag || other_flag; }
			}
			U incoming = wave64::shift_up_by_one<U>(acc, 0);              // sum since the last head, over lanes < l
			bool head_before = (bool) wave64::shift_up_by_one<int>((int) flag, 0);   // is there a head in lanes < l
			U lane_carry = head_before ? incoming : (U) (incoming + carry);
			#pragma unroll
			for (unsigned j = 0; j < E; j++) {
				if (p0 + j < end) {
					U v = before_head[j] ? (U) (value[j] + lane_carry) : value[j];
					value[j] = (U) (anchors[seg_of[j]] + v);
				}
			}
			patches.template apply_to_row<E>(row_start, lane, value);
			#pragma unroll
			for (unsigned j = 0; j < E; j++) { if (p0 + j < end) { out[p0 + j] = value[j]; } }
			U last_acc = wave64::read_last_lane(acc);
			bool any_head = wave64::ballot(flag) >> 63;   // lane 63's inclusive flag
			carry = any_head ? last_acc : (U) (carry + last_acc);
		}
	}
}

} // namespace detail

namespace detail {

/// whether an instantiation's wave path brings a chunk's input in through LDS, and the bytes of LDS a wavefront needs for it
template <unsigned UncompressedSize, unsigned CompressedSize, bool Patched> constexpr bool stages_row_inputs()
{
	return row_traits<UncompressedSize, CompressedSize>::E * CompressedSize <= (Patched ? patched_stage_row_inputs_up_to : stage_row_inputs_up_to);
}
template <unsigned UncompressedSize, unsigned CompressedSize> constexpr unsigned staged_chunk_bytes()
{
	return rows_per_chunk * row_traits<UncompressedSize, CompressedSize>::row * CompressedSize;
}

/// picks the path (wave-uniform decision): see the notes at the top of this file
template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize, typename Patches, typename Grid = launch_grid::whole>
__device__ __forceinline__ void decompress(
	uint_t<UncompressedSize>* __restrict__ decompressed, const uint_t<CompressedSize>* __restrict__ compressed_input,
	const uint_t<UncompressedSize>* __restrict__ anchor_values, uint64_t n, uint64_t L, Patches patches,
	uint8_t* provided_input_staging = nullptr,   // (the fused kernel's: 4 x staged_chunk_bytes, shared with its patch windows)
	Grid grid = Grid())   // (launch_grid.hpp: the launch's own, or a batch piece's part of it)
{
	using traits = row_traits<UncompressedSize, CompressedSize>;
	if (n == 0 || L == 0) { return; }
	// 16-byte access works whenever segments start on a lane's group boundary
	const bool vector_access = (L % traits::E == 0)
		&& memory_access::is_aligned(decompressed, 16)
		&& memory_access::is_aligned(compressed_input, traits::E * CompressedSize);
	const uint64_t rows_per_segment = (L + traits::row - 1) / traits::row;
	if (any_length_by_windows<UncompressedSize, CompressedSize>(L) && wave64::per_block() <= any_length_waves
		&& memory_access::is_aligned(decompressed, 16) && memory_access::is_aligned(compressed_input, traits::E * CompressedSize)) {
		__shared__ uint_t<UncompressedSize> bases_of_segments[any_length_waves][traits::row];
		wave_windows_of_any_segments<UncompressedSize, CompressedSize>(decompressed, compressed_input, anchor_values, n, L, patches, bases_of_segments[wave64::index_in_block()], grid);
		return;
	}
	if (vector_access && takes_wave_path<UncompressedSize, CompressedSize>(n, L)) {
		if constexpr (stages_row_inputs<UncompressedSize, CompressedSize, Patches::enabled>()) {
			// (staging exists for the 4 wavefronts of the wave path's 256-thread workgroups)
			uint8_t* staging = provided_input_staging;
			if constexpr (!Patches::enabled) {
				__shared__ __attribute__((aligned(16))) uint8_t input_staging[4][staged_chunk_bytes<UncompressedSize, CompressedSize>()];
				staging = &input_staging[0][0];
			}
			if (staging != nullptr && L % traits::row == 0 && wave64::per_block() <= 4 && memory_access::is_aligned(compressed_input, 16)) {
				wave_per_segment_with_staged_input<UncompressedSize, CompressedSize>(
					decompressed, compressed_input, anchor_values, n, L, patches,
					staging + wave64::index_in_block() * staged_chunk_bytes<UncompressedSize, CompressedSize>(), grid);
				return;
			}
			// (segments sharing rows, in chunks of exactly the staging area's size: staged too)
			if (ragged_segments_staged(UncompressedSize) && staging != nullptr && wave64::per_block() <= 4 && memory_access::is_aligned(compressed_input, 16)
				&& short_segments_share_rows<UncompressedSize, CompressedSize>(L)
				&& wave_chunk_length_of<UncompressedSize, CompressedSize>(L) == (uint64_t) rows_per_chunk * traits::row) {
				wave_rows_of_short_segments<UncompressedSize, CompressedSize, true>(
					decompressed, compressed_input, anchor_values, n, L, patches,
					staging + wave64::index_in_block() * staged_chunk_bytes<UncompressedSize, CompressedSize>(), grid);
				return;
			}
			// (ragged segments of at least a row, the chunk's bytes 16-byte aligned: staged as well, rows by the output's cache lines)
			if (ragged_segments_staged(UncompressedSize) && staging != nullptr && L >= traits::row && wave64::per_block() <= 4 && memory_access::is_aligned(compressed_input, 16)
				&& (rows_per_segment > rows_per_chunk || segments_per_wave_chunk(rows_per_segment) * L * CompressedSize + 16 <= staged_chunk_bytes<UncompressedSize, CompressedSize>())) {
				wave_per_ragged_segment_with_staged_input<UncompressedSize, CompressedSize>(
					decompressed, compressed_input, anchor_values, n, L, patches,
					staging + wave64::index_in_block() * staged_chunk_bytes<UncompressedSize, CompressedSize>(), grid);
				return;
			}
		}
		if (L % traits::row == 0) { wave_per_segment<UncompressedSize, CompressedSize>(decompressed, compressed_input, anchor_values, n, L, patches, grid); }
		else if (short_segments_share_rows<UncompressedSize, CompressedSize>(L)) { wave_rows_of_short_segments<UncompressedSize, CompressedSize>(decompressed, compressed_input, anchor_values, n, L, patches, nullptr, grid); }
		else if (L >= traits::row) { wave_per_ragged_segment_in_aligned_rows<UncompressedSize, CompressedSize>(decompressed, compressed_input, anchor_values, n, L, patches, grid); }
		else { wave_per_ragged_segment<UncompressedSize, CompressedSize>(decompressed, compressed_input, anchor_values, n, L, patches, grid); }
	}
	else if (vector_access && wave64::per_block() <= max_waves_per_block) {
		block_per_segment<UncompressedSize, CompressedSize>(decompressed, compressed_input, anchor_values, n, L, patches, grid);
	}
	else {
		segmented<IndexSize, UncompressedSize, CompressedSize>(decompressed, compressed_input, anchor_values, n, L, patches, grid);
	}
}

} // namespace detail

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
__global__ void decompress(
	uint_t<UncompressedSize>*        __restrict__  decompressed,
	const uint_t<CompressedSize>*    __restrict__  compressed_input,
	const uint_t<UncompressedSize>*  __restrict__  anchor_values,
	size_type_by_index_size<IndexSize>             length,
	size_type_by_index_size<IndexSize>             segment_length)
{
	detail::decompress<IndexSize, UncompressedSize, CompressedSize>(
		decompressed, compressed_input, anchor_values, length, segment_length, sorted_patches::none());
}

/// A piece of a batch launch (giddy/batch.hpp; launched as batch::decode_pieces<piece_body<...>>): `decompress`'s parameters out of the piece's record, its body
/// — every path of it — on the piece's part of the grid
template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
struct piece_body {
	enum : unsigned { max_threads_per_block = 1024 };
	static __device__ __forceinline__ void run(const batch::piece_view piece, const launch_grid::part grid)
	{
		detail::decompress<IndexSize, UncompressedSize, CompressedSize>(
			piece.get<uint_t<UncompressedSize>*>(0), piece.get<const uint_t<CompressedSize>*>(1), piece.get<const uint_t<UncompressedSize>*>(2),
			piece.get<size_type_by_index_size<IndexSize>>(3), piece.get<size_type_by_index_size<IndexSize>>(4), sorted_patches::none(), nullptr, grid);
	}
};

/// The same for a column at addresses aligned for its elements only (a slice [a, b) decoded into `decompressed + a`, the deltas at `compressed_input + a`), where
/// `decompress` goes element by element (17-21 % of the HBM peak): windows of the column whose rows follow the output's 1 KiB boundaries, every element finding
/// its segment (detail::wave_windows_of_any_segments, `AnyAlignment`) — for segments of up to detail::rows_per_chunk rows and workgroups of up to 4 wavefronts, which
/// is what the wrapper launches it for; anything else: the element-wise path.
template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
__global__ void decompress_at_any_alignment(
	uint_t<UncompressedSize>*        __restrict__  decompressed,
	const uint_t<CompressedSize>*    __restrict__  compressed_input,
	const uint_t<UncompressedSize>*  __restrict__  anchor_values,
	size_type_by_index_size<IndexSize>             length,
	size_type_by_index_size<IndexSize>             segment_length)
{
	using traits = detail::row_traits<UncompressedSize, CompressedSize>;
	const uint64_t n = length, L = segment_length;
	if (n == 0 || L == 0) { return; }
	if (L <= (uint64_t) detail::rows_per_chunk * traits::row && wave64::per_block() <= detail::any_length_waves) {
		__shared__ uint_t<UncompressedSize> bases_of_segments[detail::any_length_waves][traits::row];
		detail::wave_windows_of_any_segments<UncompressedSize, CompressedSize, sorted_patches::none, true>(
			decompressed, compressed_input, anchor_values, n, L, sorted_patches::none(), bases_of_segments[wave64::index_in_block()]);
		return;
	}
	detail::segmented<IndexSize, UncompressedSize, CompressedSize>(decompressed, compressed_input, anchor_values, n, L, sorted_patches::none());
}

// ---- segments longer than a window at any alignment (round 6) -------------------------------------------------------------------
//
// decompress_at_any_alignment's windows add up, for a window that begins inside a segment, that segment's deltas before the window: fine while a segment is at most a
// window (32 rows), hopeless beyond (a segment of S windows reads its input S / 2 times) — such columns went element by element (17-21 % of the HBM peak).  Here the value
// before every window comes from a PRE-PASS instead, whatever the segment length: (1) window_tail_sums — a wavefront per window adds up the window's deltas from the segment
// boundary inside it (there is at most one: L > window) or from its start; (2) bases_of_windows — a wavefront per window adds its segment's anchor and the tail sums of the
// windows between the segment's start and itself; (3) the windows path with the bases given.  The deltas are read twice: (2 C + U) / (C + U) of the bytes.
namespace detail {

/// sums of every 64 lanes' partial sums, wave-wide, in the uncompressed type (wrapping as the decoder's own sums do)
template <typename U>
__device__ __forceinline__ U wave_sum(U mine)
{
	#pragma unroll
	for (unsigned d = 32; d > 0; d >>= 1) { mine = (U) (mine + (U) __shfl_xor(mine, d, 64)); }
	return mine;
}

} // namespace detail

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
__global__ void __launch_bounds__(256) window_tail_sums(
	uint_t<UncompressedSize>*        __restrict__  tail_sums,          // one per window of detail::rows_per_chunk rows
	const uint_t<CompressedSize>*    __restrict__  compressed_input,
	size_type_by_index_size<IndexSize>             length,
	size_type_by_index_size<IndexSize>             segment_length)
{
	using traits = detail::row_traits<UncompressedSize, CompressedSize>;
	using U = typename traits::uncompressed_type;
	constexpr unsigned E = traits::E;
	constexpr uint64_t window = (uint64_t) detail::rows_per_chunk * traits::row;
	const uint64_t n = length, L = segment_length;
	const uint64_t num_windows = (n + window - 1) / window;
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) blockIdx.x * wave64::per_block() + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) gridDim.x * wave64::per_block();
	for (uint64_t c = wave; c < num_windows; c += num_waves) {
		const uint64_t start = c * window, end = start + window < n ? start + window : n;
		const uint64_t boundary = (start + L - 1) / L * L;   // the first segment start at or behind the window's
		const uint64_t from = boundary < end ? boundary : start;
		const uint64_t aligned_from = from / E * E;   // (a lane's group of E elements: those before `from` are left out)
		U sum = 0;
		for (uint64_t at = aligned_from; at < end; at += 4 * traits::row) {
			detail::row_t<UncompressedSize, CompressedSize> rows[4];
			#pragma unroll
			for (unsigned k = 0; k < 4; k++) { rows[k].template load<true>(compressed_input, at + (uint64_t) k * traits::row, end, lane); }
			#pragma unroll
			for (unsigned k = 0; k < 4; k++) {
				const uint64_t q = at + (uint64_t) k * traits::row + (uint64_t) lane * E;
				#pragma unroll
				for (unsigned j = 0; j < E; j++) { sum = (U) (sum + ((q + j >= from && q + j < end) ? (U) rows[k].deltas[j] : (U) 0)); }
			}
		}
		sum = detail::wave_sum(sum);
		if (lane == 0) { tail_sums[c] = sum; }
	}
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
__global__ void __launch_bounds__(256) bases_of_windows(
	uint_t<UncompressedSize>*        __restrict__  window_bases,       // [c]: the value before window c's first element where the window begins inside a segment (else unused)
	const uint_t<UncompressedSize>*  __restrict__  tail_sums,
	const uint_t<UncompressedSize>*  __restrict__  anchor_values,
	size_type_by_index_size<IndexSize>             length,
	size_type_by_index_size<IndexSize>             segment_length)
{
	using traits = detail::row_traits<UncompressedSize, CompressedSize>;
	using U = typename traits::uncompressed_type;
	constexpr uint64_t window = (uint64_t) detail::rows_per_chunk * traits::row;
	const uint64_t n = length, L = segment_length;
	const uint64_t num_windows = (n + window - 1) / window;
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) blockIdx.x * wave64::per_block() + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) gridDim.x * wave64::per_block();
	for (uint64_t c = wave; c < num_windows; c += num_waves) {
		const uint64_t start = c * window;
		const uint64_t segment = start / L, origin = segment * L;
		if (origin == start) { continue; }   // (wave-uniform: the window begins a segment)
		U sum = 0;
		for (uint64_t w = origin / window + lane; w < c; w += wave64::size) { sum = (U) (sum + tail_sums[w]); }
		sum = detail::wave_sum(sum);
		if (lane == 0) { window_bases[c] = (U) (anchor_values[segment] + sum); }
	}
}

/// decompress_at_any_alignment for segments of MORE than detail::rows_per_chunk rows, given the windows' bases (workgroups of up to 4 wavefronts)
template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
__global__ void decompress_windows_at_any_alignment(
	uint_t<UncompressedSize>*        __restrict__  decompressed,
	const uint_t<CompressedSize>*    __restrict__  compressed_input,
	const uint_t<UncompressedSize>*  __restrict__  anchor_values,
	size_type_by_index_size<IndexSize>             length,
	size_type_by_index_size<IndexSize>             segment_length,
	const uint_t<UncompressedSize>*  __restrict__  window_bases)
{
	using traits = detail::row_traits<UncompressedSize, CompressedSize>;
	if (length == 0 || (uint64_t) segment_length <= (uint64_t) detail::rows_per_chunk * traits::row) { return; }
	detail::wave_windows_of_long_segments<UncompressedSize, CompressedSize>(decompressed, compressed_input, anchor_values, length, segment_length, window_bases);
}

// ---- long segments: tiles chained by a decoupled look-back ---------------------------------------------------------------
//
// A segment of many rows is one dependent chain: the block path gives it to ONE workgroup (a column of a few long segments
// leaves most of the chip idle — a single 2^24-element segment ran at 8 % of the HBM peak), the parts path of round 2 read the
// deltas twice.  Here the column is cut into TILES of `chained::tile_rows` rows whatever the segment length; every tile is a
// workgroup of its own, and what a tile needs from the tiles before it — the value before its first element — travels
// through a small array of per-tile states in device memory, single pass:
//
//   1. the tile loads its deltas, scans its rows and knows its total;
//   2. it publishes the total (state AGGREGATE) — or, if it begins a segment, anchor + total right away (state PREFIX);
//   3. its first wavefront looks back over the states of the preceding tiles, 64 at a time (one per lane), adding aggregates
//      until it meets a PREFIX — never past the segment's first tile, which always publishes one;
//   4. it publishes its own PREFIX, and stores its rows.
//
// A state is one or two self-validating 8-byte words {32 bits of value, 32 bits of tag}: written and read with single agent-scope
// (memory-side) 8-byte accesses, so a reader either sees a word of this launch or not yet — no fence, no flag to order against
// (MI355X_MICROARCH.md: data-tagged granules).  The array is zeroed before the launch (kernel_wrappers/decompression/delta.hpp).
// Nothing here depends on the order in which workgroups are dispatched: a tile that does not see its predecessors' states
// within `chained::patience` polls adds up the deltas before it itself (from the segment's start — correct, just slow), so the
// launch completes under any schedule; with the dispatcher's in-order habit it never happens.

namespace chained {

enum : unsigned {
	aggregate = 1, prefix = 2,
	patience = 1u << 16,       // polls (~1 us each) that bring nothing new before a tile stops waiting
	lookback_windows = 1,   // predecessors read per poll, in 64s
	rows_of_the_second_pass_together = 2,   // rows of the second pass the instruction scheduler may interleave
};

/// Rows a wavefront takes of its tile: as many as it can keep the DELTAS of in registers between the two passes over them (totals
/// first; prefixes and stores once the tile's base is known) — 32 rows of up to 4 bytes per lane, 16 of 8, 8 of 16.  Big tiles
/// matter: a tile looks back over 64 predecessors per poll of ~1 us, so the look-back keeps up with at most ~64 tiles per us chip-wide;
/// tiles of 32 rows (the first version) start at 150 per us at full speed, fell ever further behind the nearest published prefix
/// and ran at 35 % of the HBM peak; tiles of 128 rows start at 38 per us.
template <unsigned UncompressedSize, unsigned CompressedSize>
constexpr unsigned rows_per_wave()
{
	return detail::row_traits<UncompressedSize, CompressedSize>::E * CompressedSize <= 4 ? 32 : (detail::row_traits<UncompressedSize, CompressedSize>::E * CompressedSize <= 8 ? 16 : 8);
}
/// A tile is a workgroup of `Waves` wavefronts (4, 8 or 16 — the wrapper picks by the number of tiles in a segment: the longer the chain,
/// the bigger the tile.  Measured, <4,8,2> at 2^28 elements, segments of 2^16 / 2^20 / 2^24 / 2^28: 4 wavefronts 75 / 58 / 47 / 46 % of
/// the HBM peak, 8: 76 / 77 / 65 / 65 %, 16: 68 / 74 / 67 / 66 % — every tile in flight polls its predecessors' states through the memory
/// side, and it is the number of tiles started per microsecond that the look-back has to keep up with).
template <unsigned UncompressedSize, unsigned CompressedSize, unsigned Waves>
constexpr uint64_t tile_elements() { return (uint64_t) Waves * rows_per_wave<UncompressedSize, CompressedSize>() * detail::row_traits<UncompressedSize, CompressedSize>::row; }

/// wavefronts per tile for segments of `segment_length` (0: these segments are not whole tiles of any size — not for this path)
template <unsigned UncompressedSize, unsigned CompressedSize>
__host__ __device__ inline unsigned waves_per_tile_for(uint64_t length, uint64_t segment_length)
{
	const bool one_segment = segment_length >= length;
	const uint64_t t16 = tile_elements<UncompressedSize, CompressedSize, 16>(), t8 = tile_elements<UncompressedSize, CompressedSize, 8>(), t4 = tile_elements<UncompressedSize, CompressedSize, 4>();
	const uint64_t span = one_segment ? length : segment_length;
	if ((one_segment || segment_length % t16 == 0) && span / t16 >= 128) { return 16; }
	if ((one_segment || segment_length % t8 == 0) && span / t8 >= 8) { return 8; }
	if (one_segment || segment_length % t4 == 0) { return 4; }
	return 0;
}

/// a tile's state: `words` 8-byte granules, each {value bits, tag}
template <unsigned UncompressedSize>
struct tile_state {
	enum : unsigned { words = UncompressedSize > 4 ? 2 : 1 };
	unsigned long long word[2];
};

template <unsigned UncompressedSize>
__device__ __forceinline__ void publish(tile_state<UncompressedSize>* state, uint_t<UncompressedSize> value, unsigned tag)
{
	const uint64_t v = (uint64_t) value;
	__hip_atomic_store(&state->word[0], ((unsigned long long) tag << 32) | (uint32_t) v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	if (UncompressedSize > 4) { __hip_atomic_store(&state->word[1], ((unsigned long long) tag << 32) | (uint32_t) (v >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
}

/// tag (0: not there yet — or the two words disagree, i.e. the state is being rewritten) and value
template <unsigned UncompressedSize>
__device__ __forceinline__ unsigned read(const tile_state<UncompressedSize>* state, uint_t<UncompressedSize>& value)
{
	const unsigned long long lo = __hip_atomic_load(&state->word[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	unsigned tag = (unsigned) (lo >> 32);
	uint64_t v = (uint32_t) lo;
	if (UncompressedSize > 4) {
		const unsigned long long hi = __hip_atomic_load(&state->word[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if ((unsigned) (hi >> 32) != tag) { tag = 0; }
		v |= (uint64_t) (uint32_t) hi << 32;
	}
	value = (uint_t<UncompressedSize>) v;
	return tag;
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize, unsigned Waves>
__global__ void __launch_bounds__(Waves * 64) decompress(
	uint_t<UncompressedSize>*        __restrict__  decompressed,
	const uint_t<CompressedSize>*    __restrict__  compressed_input,
	const uint_t<UncompressedSize>*  __restrict__  anchor_values,
	size_type_by_index_size<IndexSize>             length,
	size_type_by_index_size<IndexSize>             segment_length,
	tile_state<UncompressedSize>*    __restrict__  states)
{
	using traits = detail::row_traits<UncompressedSize, CompressedSize>;
	using U = typename traits::uncompressed_type;
	using C = typename traits::compressed_type;
	using row_type = detail::row_t<UncompressedSize, CompressedSize>;
	constexpr unsigned R = rows_per_wave<UncompressedSize, CompressedSize>();
	constexpr unsigned waves_per_tile = Waves;
	constexpr uint64_t tile = tile_elements<UncompressedSize, CompressedSize, Waves>();
	__shared__ U wave_totals[waves_per_tile];
	__shared__ U base_of_tile;
	const unsigned lane = wave64::lane_id(), w = wave64::index_in_block();
	const uint64_t n = length, L = segment_length;
	const uint64_t tiles = (n + tile - 1) / tile;
	const uint64_t tiles_per_segment = L >= n ? tiles : L / tile;   // (the wrapper sends only whole-tile segments, or one segment, here)
	sorted_patches::none no_patches;

	for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
		const uint64_t tile_start = t * tile, tile_end = tile_start + tile < n ? tile_start + tile : n;
		const uint64_t segment = t / tiles_per_segment, first_tile_of_segment = segment * tiles_per_segment;
		const bool begins_segment = t == first_tile_of_segment;

		// 1. this wavefront's rows: the deltas stay in registers; first pass, the rows' totals
		memory_access::group<C, traits::E> deltas[R];
		const uint64_t my_start = tile_start + (uint64_t) w * R * traits::row;
		// (a whole tile — every tile but the column's last — loads and stores without a bound in sight: with the bounded forms in the
		// same code the compiler keeps both alive for all the rows, 200 registers)
		const bool whole = tile_end - tile_start == tile;
		if (whole) {
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { deltas[k] = memory_access::load_group<C, traits::E>(compressed_input + my_start + (uint64_t) k * traits::row + (uint64_t) lane * traits::E); }
		}
		U anchor = 0;
		if (begins_segment) { anchor = anchor_values[segment]; }
		U my_total = 0;
		if (whole) {
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				typename traits::scan_type mine = 0;
				#pragma unroll
				for (unsigned j = 0; j < traits::E; j++) { mine += deltas[k][j]; }
				my_total += (U) mine;
			}
		}
		else {
			// (the column's last tile: row by row, the deltas read again in the second pass — no registers held for it)
			#pragma unroll 1
			for (unsigned k = 0; k < R; k++) {
				row_type row;
				row.load(compressed_input, my_start + (uint64_t) k * traits::row, tile_end, lane);
				#pragma unroll
				for (unsigned j = 0; j < traits::E; j++) { my_total += (U) row.deltas[j]; }
			}
		}
		for (unsigned d = 32; d > 0; d >>= 1) { my_total += (U) __shfl_xor(my_total, d, 64); }   // (one reduction for all the rows)
		if (lane == 0) { wave_totals[w] = my_total; }
		__syncthreads();
		U before_me = 0, tile_total = 0;
		#pragma unroll
		for (unsigned v = 0; v < waves_per_tile; v++) { const U x = wave_totals[v]; tile_total += x; if (v < w) { before_me += x; } }

		// 2. the tile's total for the tiles behind it — by the LAST wavefront, so that the first one's polling loads do not queue behind
		// this store's acknowledgement (vector-memory operations of a wavefront return in order)
		if (w == waves_per_tile - 1 && lane == 0 && !begins_segment) { publish(states + t, tile_total, aggregate); }

		// 3. the value before the tile's first element
		if (w == 0) {
			U base;
			if (begins_segment) { base = anchor; }
			else {
				// Look back: lane l of sub-window j reads the state of tile nearest - 1 - (64 j + l), all of a poll's reads in flight together.  The
				// states are taken in order of distance: aggregates are added up as far as they are there, a PREFIX ends the walk; whatever
				// was added is kept (`nearest` moves back), so a poll that meets a tile without a state yet still makes progress.
				base = 0;
				uint64_t nearest = t;   // tiles [nearest, t) are accounted for in `base`
				unsigned idle_polls = 0;
				for (bool found = false; !found; ) {
					U value[lookback_windows]; unsigned tag[lookback_windows]; bool looks[lookback_windows];
					#pragma unroll
					for (unsigned j = 0; j < lookback_windows; j++) {
						looks[j] = nearest - first_tile_of_segment > 64 * j + lane;   // (never before the segment's first tile)
						value[j] = 0; tag[j] = 0;
						if (looks[j]) { tag[j] = read(states + (nearest - 1 - (64 * j + lane)), value[j]); }
					}
					uint64_t advanced = 0;
					#pragma unroll
					for (unsigned j = 0; j < lookback_windows && !found; j++) {
						const uint64_t there = wave64::ballot(!looks[j] || tag[j] != 0), is_prefix = wave64::ballot(looks[j] && tag[j] == prefix);
						const unsigned first_missing = there == ~0ull ? 64u : (unsigned) __builtin_ctzll(~there);
						const unsigned first_prefix = is_prefix == 0 ? 64u : (unsigned) __builtin_ctzll(is_prefix);
						const bool ends_here = first_prefix < first_missing;
						const unsigned take = ends_here ? first_prefix + 1 : first_missing;   // states 0 .. take - 1 of this sub-window
						U mine = (looks[j] && lane < take) ? value[j] : (U) 0;
						for (unsigned d = 32; d > 0; d >>= 1) { mine += (U) __shfl_xor(mine, d, 64); }
						base += mine;
						if (ends_here) { found = true; break; }
						advanced += take;
						if (first_missing < 64u) { break; }   // a tile without a state yet: poll again from there
					}
					if (found) { break; }
					nearest -= advanced;
					if (advanced == 0) {
						if (++idle_polls == patience) {
							// out of patience: the deltas from the segment's start to the tiles already accounted for, added up here
							const uint64_t from = first_tile_of_segment * tile, to = nearest * tile;
							U sum = 0;
							for (uint64_t i = from + lane; i < to; i += 64) { sum += (U) compressed_input[i]; }
							for (unsigned d = 32; d > 0; d >>= 1) { sum += (U) __shfl_xor(sum, d, 64); }
							base += (U) (anchor_values[segment] + sum);
							break;
						}
						__builtin_amdgcn_s_sleep(4);
					}
					else { idle_polls = 0; }
				}
			}
			if (lane == 0) { base_of_tile = base; }
		}
		__syncthreads();
		// 4. everything up to this tile's end, for the tiles behind it (again not by the polling wavefront), and the rows
		if (w == waves_per_tile - 1 && lane == 0) { publish(states + t, (U) (base_of_tile + tile_total), prefix); }
		U base = (U) (base_of_tile + before_me);
		if (whole) {
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				row_type row;
				row.deltas = deltas[k];
				row.scan();
				memory_access::group<U, traits::E> values;
				#pragma unroll
				for (unsigned j = 0; j < traits::E; j++) { values[j] = (U) (base + (U) row.prefix[j]); }
				memory_access::store_group<memory_access::wide_output_policy<sizeof(U)>(GIDDY_DELTA_STORE_POLICY)>(
					decompressed + my_start + (uint64_t) k * traits::row + (uint64_t) lane * traits::E, values);
				base += (U) row.total;
				if (k % rows_of_the_second_pass_together == rows_of_the_second_pass_together - 1) { __builtin_amdgcn_sched_barrier(0); }   // (left alone the scheduler scans all the rows first)
			}
		}
		else {
			#pragma unroll 1
			for (unsigned k = 0; k < R; k++) {
				row_type row;
				row.load(compressed_input, my_start + (uint64_t) k * traits::row, tile_end, lane);
				row.scan();
				row.store(decompressed, my_start + (uint64_t) k * traits::row, tile_end, lane, base, no_patches);
				base += (U) row.total;
			}
		}
		__syncthreads();   // (wave_totals and base_of_tile are rewritten by the workgroup's next tile)
	}
}

} // namespace chained

namespace patched {

/// Delta decompression and patching in one pass: the result of `delta::decompress` followed by
/// `data_layout::scatter::scatter(decompressed, patch_values, patch_positions, num_patches)` — for ASCENDING
/// patch positions (equal positions: the last one wins) — without the second launch or its read-modify-write.
template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize, bool ManyPatches = false>
__global__ void decompress(
	uint_t<UncompressedSize>*        __restrict__  decompressed,
	const uint_t<CompressedSize>*    __restrict__  compressed_input,
	const uint_t<UncompressedSize>*  __restrict__  anchor_values,
	const uint_t<UncompressedSize>*  __restrict__  patch_values,
	const uint_t<IndexSize>*         __restrict__  patch_positions,
	size_type_by_index_size<IndexSize>             length,
	size_type_by_index_size<IndexSize>             segment_length,
	size_type_by_index_size<IndexSize>             num_patches)
{
	// LDS: patch windows for up to 16 wavefronts, and input staging for the 4 of the wave path's workgroups.  The two never meet in
	// full — staging is for workgroups of at most 4 wavefronts — so the windows of wavefronts 4..15 overlay the staging area.
	using value_type = uint_t<UncompressedSize>;
	using more_windows = sorted_patches::window_storage<IndexSize, value_type, 12>;
	constexpr bool stages = detail::stages_row_inputs<UncompressedSize, CompressedSize, true>();
	constexpr unsigned staging_bytes = stages ? 4 * detail::staged_chunk_bytes<UncompressedSize, CompressedSize>() : 0;
	__shared__ sorted_patches::window_storage<IndexSize, value_type, 4> first_windows;
	__shared__ sorted_patches::row_storage<value_type, ManyPatches ? 4 : 0, 16 / sizeof(value_type), IndexSize, 1> patch_rows;   // (one page of patches: sorted_patches.hpp)
	__shared__ __attribute__((aligned(16))) uint8_t overlay[staging_bytes > sizeof(more_windows) ? staging_bytes : sizeof(more_windows)];
	const unsigned w = wave64::index_in_block() % 16;
	more_windows& later_windows = *reinterpret_cast<more_windows*>(overlay);
	detail::decompress<IndexSize, UncompressedSize, CompressedSize>(
		decompressed, compressed_input, anchor_values, length, segment_length,
		sorted_patches::cursor_t<IndexSize, value_type, ManyPatches, 16 / sizeof(value_type), 1>(patch_values, patch_positions, num_patches, length,
			w < 4 ? first_windows.positions[w] : later_windows.positions[w - 4], w < 4 ? first_windows.values[w] : later_windows.values[w - 4]).with_rows(patch_rows),
		stages ? overlay : nullptr);
}

} // namespace patched

template <unsigned IndexSize, unsigned UncompressedSize, unsigned CompressedSize>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;

	launch_config_resolution_params_t(
		device::properties_t            device_properties_,
		device_function::attributes_t   kernel_function_attributes_,
		size_t                          data_length,
		size_t                          segment_length,
		optional<shared_memory_size_t>  dynamic_shared_mem_limit = nullopt)
	: parent(device_properties_, kernel_function_attributes_, dynamic_shared_mem_limit)
	{
		using traits = detail::row_traits<UncompressedSize, CompressedSize>;
		if (segment_length == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
		auto num_segments = util::div_rounding_up(data_length, segment_length);
		bool vector_access = segment_length % traits::E == 0;
		size_t rows_per_segment = util::div_rounding_up(segment_length, (size_t) traits::row);
		bool wave_path = vector_access && detail::takes_wave_path<UncompressedSize, CompressedSize>(data_length, segment_length);
		bool block_path = vector_access && !wave_path;
		if (block_path) {
			// one workgroup per segment (segments at grid stride), as many wavefronts as one sweep needs
			size_t waves = std::min<size_t>(detail::max_waves_per_block,
				util::div_rounding_up(rows_per_segment, (size_t) detail::rows_per_wave_per_sweep));
			grid_construction_resolution = block;
			length = num_segments;
			if (waves <= 2) {   // (segments of 8 or 16 rows: 4 or 2 of them to a workgroup of 4 wavefronts — block_per_segment)
				length = util::div_rounding_up(num_segments, (size_t) (4 / waves));
				waves = 4;
			}
			block_resolution_constraints.max_threads_per_block = (grid_block_dimension_t) (waves * warp_size);
			upper_limits.threads_in_block = (grid_block_dimension_t) (waves * warp_size);
		}
		else {
			// one wavefront per segment (or per batch of short segments), at grid stride
			grid_construction_resolution = warp;
			size_t per_chunk = segment_length >= 256 ? 1 : util::div_rounding_up((size_t) 256, segment_length);
			length = wave_path ? util::div_rounding_up(data_length, (size_t) detail::wave_chunk_length_of<UncompressedSize, CompressedSize>(segment_length))
			       : detail::any_length_by_windows<UncompressedSize, CompressedSize>(segment_length) ? util::div_rounding_up(data_length, (size_t) detail::rows_per_chunk * traits::row)
			       : util::div_rounding_up(num_segments, per_chunk);
			upper_limits.threads_in_block = detail::wave_path_threads;
			// 1-byte deltas of 8-byte elements: 4 resident workgroups per CU are 7 points faster than what fits (measured per instantiation, tools/occupancy_sweep.py)
			if (wave_path && UncompressedSize == 8 && CompressedSize == 1) { resident_blocks_per_multiprocessor_cap = 4u; }
		}
		serialization_option = auto_maximized;
		keep_gpu_busy_factor = 1024;
		quanta.threads_in_block = warp_size;
	}

	launch_config_resolution_params_t(
		device::properties_t device_properties_, size_t data_length, size_t segment_length,
		optional<shared_memory_size_t> dynamic_shared_mem_limit = nullopt)
	: launch_config_resolution_params_t(device_properties_,
		device_function_t(decompress<IndexSize, UncompressedSize, CompressedSize>).attributes(),
		data_length, segment_length, dynamic_shared_mem_limit) { }
};

} // namespace delta
} // namespace decompression
} // namespace kernels
} // namespace cuda
