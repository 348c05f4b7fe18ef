// Discard-zero-bytes (null suppression), VARIABLE width, for CDNA4.
//
// Element i keeps only its n_i low-order bytes, stored back to back (little-endian) in
// compressed_data; n_i = min_represented_element_size + f_i, where f_i is a bits_per_element_size-bit
// field, fields packed LSB-first into containers of ElementSizesContainerSize bytes,
// floor(container bits / field bits) per container, never straddling (high slack bits unused);
// position_anchors[s] is the byte offset of element s*segment_length.
//
// Same kernel template name, parameter list and element semantics as the reference's
// src/kernels/decompression/discard_zero_bytes_variable.cuh:381-439 (get_element_size :44-58,
// pad_high_bytes :196-213, warp body :241-330) — except that every tail length is handled (the
// reference's last-warp test is only right when length % segment_length is 0 or <= 32, quirk a8).
//
// Execution.  The reference gives a 32-lane warp a whole segment and moves one element per lane per
// iteration: integer div+mod, a 5-step shuffle scan, a byte-wise unaligned memcpy from global
// memory and a broadcast, per 32 elements.  Here a WAVEFRONT owns a segment (segments at grid
// stride) and works in rows of 64*E elements, E = 16/UncompressedSize consecutive elements per lane:
//   1. size fields: the lane's position within its container by a reciprocal multiply, then E shifts/masks;
//   2. byte offsets: in-lane running sum + one 64-lane DPP prefix sum (row byte counts < 2^11);
//   3. the row's compressed bytes — one contiguous range of <= 1 KiB — are staged into LDS with
//      16-byte-per-lane COALESCED loads of exactly the 16-byte chunks that overlap the range (so the
//      kernel never touches a chunk holding no valid byte: nothing is read past the buffer's end);
//   4. each element is cut out of LDS with aligned 32-bit reads + v_alignbyte and masked to n_i bytes through
//      a 9-entry mask table kept in LDS;
//   5. one 16-byte store per lane, contiguous across the wavefront.
// Rows are processed two at a time and software-pipelined two groups deep (see the kernel body), so
// HBM round trips overlap instead of adding up.  No __syncthreads; wavefronts are independent.
//
// The WIDE path (detail::decompress_wide; 1-, 2- and 4-bit fields, 4- and 8-byte elements, segments of at least 8
// rows): a lane takes a whole 32-bit word of size fields, so ONE prefix sum covers up to 1024 elements, the
// segment's whole byte range is staged with two to five loads, and a row costs two ds_bpermute and a SWAR prefix
// instead of a size-word load, a scan and a chunk-address computation.  C5 (2^28 x u64, 2-bit fields): 67-72 -> 74-77 %
// of the HBM peak across boxes.
//
// HBM bytes per element: mean n_i + field bits/8 + UncompressedSize + anchor/segment_length
// (10.76 B for 1..4-byte sizes into 64 bits, config C5).
#pragma once
#include <hip/hip_runtime.h>

#include "giddy/common.hpp"
#include "giddy/kernels/launch_config_resolution_params.hpp"
#include "giddy/on_device/wave64.hpp"
#include "giddy/util/math.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace discard_zero_bytes {
namespace variable_width {

using element_size_t = native_word_t;

// build-time tuning knob (default measured on MI355X)
#ifndef GIDDY_DZBV_ROWS_IN_FLIGHT
#define GIDDY_DZBV_ROWS_IN_FLIGHT 2
#endif

// the wide path (see detail::decompress_wide): 1 = on where it applies
#ifndef GIDDY_DZBV_WIDE
#define GIDDY_DZBV_WIDE 1
#endif
// 1 KiB pieces of a super-row's bytes requested before the first is written to LDS (measured: 2 >= 3 > 4, which costs a wavefront of occupancy)
#ifndef GIDDY_DZBV_WIDE_PIECES_IN_FLIGHT
#define GIDDY_DZBV_WIDE_PIECES_IN_FLIGHT 2
#endif
// wavefronts per SIMD the register allocator must leave room for; 0 = its own choice (65-67 vector registers, 7 wavefronts:
// measured 1-3 points FASTER than forcing 64 registers, which turns scalar spills into extra vector work)
#ifndef GIDDY_DZBV_WAVES_PER_SIMD
#define GIDDY_DZBV_WAVES_PER_SIMD 0
#endif
#if GIDDY_DZBV_WAVES_PER_SIMD > 0
#define GIDDY_DZBV_OCCUPANCY __attribute__((amdgpu_waves_per_eu(GIDDY_DZBV_WAVES_PER_SIMD)))
#else
#define GIDDY_DZBV_OCCUPANCY
#endif
#ifndef GIDDY_DZBV_SGPRS
#define GIDDY_DZBV_SGPRS 94
#endif

namespace detail {

enum : unsigned {
	max_waves_per_block = 4,          // the launch parameters cap the workgroup at 256 threads
	rows_in_flight = GIDDY_DZBV_ROWS_IN_FLIGHT,
	rows_per_chunk = 8,               // a wavefront batches short segments up to this many rows per outer-loop trip
	staging_bytes = 1024 + 16 + 16,   // a row's bytes + head misalignment + over-read slack for the aligned dword reads
	staging_words = staging_bytes / 4,
};

} // namespace detail

namespace detail {

/// byte masks by element size, 0..8 bytes — kept in LDS so that masking costs one LDS read instead of ~8 VALU ops
struct size_masks { uint64_t mask[9]; };

__device__ __forceinline__ void fill(size_masks& table)
{
	if (threadIdx.x < 9) { table.mask[threadIdx.x] = threadIdx.x == 8 ? ~(uint64_t) 0 : ((((uint64_t) 1) << (8 * threadIdx.x)) - 1); }
	__syncthreads();
}

/// low `size` bytes of the 8 bytes starting `shift_bytes` into the three aligned words (w0, w1, w2)
template <unsigned UncompressedSize>
__device__ __forceinline__ uint64_t cut(uint32_t w0, uint32_t w1, uint32_t w2, unsigned shift_bytes, unsigned size, const size_masks& masks)
{
	// v_alignbyte_b32: bytes [shift, shift+4) of the 8-byte pair {hi, lo}
	const uint64_t mask = masks.mask[size];
	uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, shift_bytes);
	if (UncompressedSize <= 4) { return lo & (uint32_t) mask; }
	uint32_t hi = __builtin_amdgcn_alignbyte(w2, w1, shift_bytes);
	return (((uint64_t) hi << 32) | lo) & mask;
}

/// Fast path (all R rows inside the segment, 16-byte aligned output, a lane's E size fields share a
/// container): the lane's E fields of row k, already shifted down to bit 0.  The group's first element
/// is field `field_base` (< per) of container `container_base` — both wave-uniform; the per-lane
/// division by `per` is a multiplication by its precomputed reciprocal.
template <unsigned UncompressedSize, unsigned ElementSizesContainerSize>
__device__ __forceinline__ void load_size_fields(
	uint_t<ElementSizesContainerSize> (&fields)[rows_in_flight],
	const uint_t<ElementSizesContainerSize>* __restrict__ packed_element_sizes,
	unsigned lane, uint64_t container_base, unsigned field_base, unsigned bits, unsigned per, unsigned per_reciprocal)
{
	constexpr unsigned E = 16 / UncompressedSize;
	#pragma unroll
	for (unsigned k = 0; k < rows_in_flight; k++) {
		// floor(index / per) == mulhi(index, ceil(2^32 / per)) exactly while index * per < 2^32 (index < 2^13, per <= 64)
		const unsigned index = field_base + k * 64 * E + lane * E;
		const unsigned q = per == 1 ? index : __umulhi(index, per_reciprocal), field = index - q * per;
		fields[k] = packed_element_sizes[container_base + q] >> (field * bits);
	}
}

template <unsigned UncompressedSize, typename Container>
__device__ __forceinline__ unsigned size_of(Container fields, unsigned j, unsigned min_size, unsigned bits, Container field_mask)
{
	unsigned size = min_size + (unsigned) ((fields >> (j * bits)) & field_mask);
	return size > UncompressedSize ? UncompressedSize : size;   // precondition n_i <= sizeof(element); never read more
}

/// where a wavefront is in the compressed byte stream (all wave-uniform)
template <unsigned IndexSize>
struct segment_cursor {
	uint64_t bytes_start;                  // byte offset of the next row's first element (wave-uniform)
	uint64_t first_anchor;                 // the anchor of the chunk's first segment (wave-uniform) ...
	uint32_t later_anchors;                // ... and, PER LANE, how far beyond it the chunk's l-th segment begins: all of a chunk's
	                                       // anchors arrive with one load, ahead of the rows, instead of one scalar load per segment
	uint32_t next_segment_in_chunk;
	uint32_t rows_left_in_segment;

	__device__ __forceinline__ void start_at(const uint_t<IndexSize>* __restrict__ position_anchors, uint64_t segment, uint64_t segment_length, uint64_t length,
		unsigned segments_per_chunk, unsigned lane)
	{
		const bool have = lane < segments_per_chunk && (segment + lane) * segment_length < length;
		const uint64_t mine = have ? (uint64_t) position_anchors[segment + lane] : 0;
		first_anchor = wave64::read_first_lane(mine);
		later_anchors = (uint32_t) (mine - first_anchor);   // (a chunk is at most rows_per_chunk rows: kilobytes)
		next_segment_in_chunk = 0; rows_left_in_segment = 0; bytes_start = 0;
	}
	__device__ __forceinline__ void begin_segment(uint32_t rows_per_segment)
	{
		bytes_start = first_anchor + wave64::read_lane(later_anchors, next_segment_in_chunk);
		next_segment_in_chunk++;
		rows_left_in_segment = rows_per_segment;
	}
	/// call before every row: moves on to the next segment's anchor when the current segment is used up
	__device__ __forceinline__ void before_row(uint32_t rows_per_segment)
	{
		if (rows_left_in_segment == 0) { begin_segment(rows_per_segment); }
		rows_left_in_segment--;
	}
};

/// segments a wavefront takes per trip around its outer loop: short segments are batched up to `rows_per_chunk` rows
__host__ __device__ inline uint64_t segments_per_wave_chunk(uint64_t segment_length, unsigned row)
{
	if (segment_length % row != 0) { return 1; }
	const uint64_t rows_per_segment = segment_length / row;
	return rows_per_segment >= rows_per_chunk ? 1 : rows_per_chunk / (unsigned) rows_per_segment;   // (no 64-bit division: a wavefront's prologue is short)
}

/// wave-uniform a / b without the ~100-instruction 64-bit division when both fit 32 bits
__device__ __forceinline__ uint64_t divide(uint64_t a, unsigned b)
{
	return (a >> 32) == 0 ? (uint64_t) ((uint32_t) a / b) : a / b;
}

/// Everything the fast path knows about a group of R full rows
template <unsigned UncompressedSize, unsigned ElementSizesContainerSize>
struct row_group {
	using container_type = uint_t<ElementSizesContainerSize>;
	using chunk_t = typename memory_access::chunk<16>::type;
	enum : unsigned { E = 16 / UncompressedSize, R = rows_in_flight };
	container_type fields[R];     // the lane's E size fields of each row, shifted down to bit 0
	unsigned       offset[R];     // byte offset of the lane's first element within the row's staging area
	unsigned       num_chunks[R]; // 16-byte chunks overlapping the row's byte range (<= 65)
	uintptr_t      first_chunk[R];
	chunk_t        chunks[R];     // this lane's chunk of each row, in flight from HBM

	/// byte offsets (DPP prefix sums) and chunk ranges of the next R rows; a row that starts an anchored segment
	/// takes its byte position from the anchor.  `cursor` (wave-uniform) is advanced.
	template <bool SpansSegments, unsigned IndexSize>
	__device__ __forceinline__ void locate(const uint8_t* __restrict__ compressed_data, uint32_t rows_per_segment,
		segment_cursor<IndexSize>& cursor, unsigned min_size, unsigned bits)
	{
		const container_type field_mask = bits >= 8 * ElementSizesContainerSize ? ~(container_type) 0 : (((container_type) 1 << bits) - 1);
		#pragma unroll
		for (unsigned k = 0; k < R; k++) {
			if (SpansSegments) { cursor.before_row(rows_per_segment); }
			unsigned lane_bytes = 0;
			#pragma unroll
			for (unsigned j = 0; j < E; j++) { lane_bytes += size_of<UncompressedSize>(fields[k], j, min_size, bits, field_mask); }
			const unsigned inclusive = wave64::inclusive_sum(lane_bytes);
			const unsigned row_bytes = wave64::read_last_lane(inclusive);
			const uintptr_t address = reinterpret_cast<uintptr_t>(compressed_data) + cursor.bytes_start;
			first_chunk[k] = address & ~(uintptr_t) 15;
			const unsigned head = (unsigned) (address - first_chunk[k]);
			num_chunks[k] = row_bytes == 0 ? 0 : (head + row_bytes + 15) / 16;
			offset[k] = inclusive - lane_bytes + head;
			cursor.bytes_start += row_bytes;
		}
	}
	/// request the rows' bytes: exactly the 16-byte chunks that overlap them.  Every lane loads — lanes past
	/// the last chunk re-read it (same cache line) — so there is no divergent branch around the loads.
	__device__ __forceinline__ void request(unsigned lane, const void* any_valid_address)
	{
		#pragma unroll
		for (unsigned k = 0; k < R; k++) {
			const unsigned last = num_chunks[k] == 0 ? 0 : num_chunks[k] - 1;
			const unsigned mine = lane < last ? lane : (last < 64 ? last : 63);
			const uintptr_t base = num_chunks[k] == 0 ? (reinterpret_cast<uintptr_t>(any_valid_address) & ~(uintptr_t) 15) : first_chunk[k];
			chunks[k] = memory_access::load_global<chunk_t>(base + 16 * (uintptr_t) mine);
		}
	}
	__device__ __forceinline__ void stage_into(uint32_t* stage, unsigned lane) const
	{
		#pragma unroll
		for (unsigned k = 0; k < R; k++) {
			chunk_t* row_stage = reinterpret_cast<chunk_t*>(stage + k * staging_words);
			row_stage[lane] = chunks[k];   // lanes past the last chunk write bytes nobody reads
			if (num_chunks[k] > 64) {      // a 1024-byte row that starts unaligned: one more chunk (rare)
				if (lane == 0) { row_stage[64] = memory_access::load_global<chunk_t>(first_chunk[k] + 16 * (uintptr_t) 64); }
			}
		}
		wave64::lds_fence();
	}
	__device__ __forceinline__ void extract_and_store(uint_t<UncompressedSize>* __restrict__ decompressed, const uint32_t* stage,
		unsigned lane, uint64_t rows_start, unsigned min_size, unsigned bits, const size_masks& masks) const
	{
		using U = uint_t<UncompressedSize>;
		const container_type field_mask = bits >= 8 * ElementSizesContainerSize ? ~(container_type) 0 : (((container_type) 1 << bits) - 1);
		#pragma unroll
		for (unsigned k = 0; k < R; k++) {
			const uint64_t i0 = rows_start + (uint64_t) k * 64 * E + (uint64_t) lane * E;
			const uint32_t* row_stage = stage + k * staging_words;
			memory_access::group<U, E> values;
			unsigned at = offset[k];
			#pragma unroll
			for (unsigned j = 0; j < E; j++) {
				const unsigned word = at >> 2, size = size_of<UncompressedSize>(fields[k], j, min_size, bits, field_mask);
				// unconditional aligned reads: the staging area has slack, bytes beyond the element are masked off
				values[j] = (U) cut<UncompressedSize>(row_stage[word], row_stage[word + 1], UncompressedSize == 8 ? row_stage[word + 2] : 0u, at & 3, size, masks);
				at += size;
			}
			memory_access::store_group(decompressed + i0, values);
		}
		wave64::lds_fence();   // the staging area may be rewritten after this
	}
};

/// R rows of one segment, starting at element rows_start.  Full: all R rows lie inside the segment and
/// the output is 16-byte aligned.  Packed: a lane's E size fields never straddle a container.
template <bool Full, bool Packed, unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
__device__ __forceinline__ uint64_t decompress_rows(
	uint_t<UncompressedSize>*  __restrict__  decompressed,
	const uint8_t*             __restrict__  compressed_data,
	const uint_t<ElementSizesContainerSize>* __restrict__ packed_element_sizes,
	uint32_t* stage, unsigned lane, uint64_t rows_start, uint64_t segment_end, uint64_t bytes_start,
	unsigned min_size, unsigned bits, unsigned per, const size_masks& masks)
{
	using namespace memory_access;
	using U = uint_t<UncompressedSize>;
	using container_type = uint_t<ElementSizesContainerSize>;
	using chunk_t = typename chunk<16>::type;
	constexpr unsigned E = 16 / UncompressedSize;
	constexpr unsigned row = 64 * E;
	constexpr unsigned R = rows_in_flight;
	const container_type field_mask = bits >= 8 * ElementSizesContainerSize ? ~(container_type) 0 : (((container_type) 1 << bits) - 1);

	unsigned sizes[R][E]; unsigned offset[R]; unsigned row_bytes[R];
	// 1. element sizes of R rows
	#pragma unroll
	for (unsigned k = 0; k < R; k++) {
		const uint64_t i0 = rows_start + (uint64_t) k * row + (uint64_t) lane * E;
		uint64_t container_index; unsigned field;
		if (IndexSize <= 4 || (i0 >> 32) == 0) { uint32_t q = (uint32_t) i0 / per; container_index = q; field = (uint32_t) i0 - q * per; }
		else { container_index = i0 / per; field = (unsigned) (i0 - container_index * per); }
		unsigned lane_bytes = 0;
		if (Full && Packed) {
			container_type fields = packed_element_sizes[container_index] >> (field * bits);
			#pragma unroll
			for (unsigned j = 0; j < E; j++) {
				unsigned size = min_size + (unsigned) ((fields >> (j * bits)) & field_mask);
				if (size > UncompressedSize) { size = UncompressedSize; }   // precondition n_i <= sizeof(element); never read more
				sizes[k][j] = size; lane_bytes += size;
			}
		}
		else {
			container_type container = (i0 < segment_end) ? packed_element_sizes[container_index] : (container_type) 0;
			#pragma unroll
			for (unsigned j = 0; j < E; j++) {
				unsigned size = 0;
				if (i0 + j < segment_end) {
					size = min_size + (unsigned) ((container >> (field * bits)) & field_mask);
					if (size > UncompressedSize) { size = UncompressedSize; }
				}
				sizes[k][j] = size; lane_bytes += size;
				if (++field == per) { field = 0; container_index++; if (j + 1 < E && i0 + j + 1 < segment_end) { container = packed_element_sizes[container_index]; } }
			}
		}
		offset[k] = lane_bytes;
	}
	// 2. byte offsets within each row (one DPP prefix sum per row; row byte counts < 2^11)
	#pragma unroll
	for (unsigned k = 0; k < R; k++) {
		const unsigned inclusive = wave64::inclusive_sum(offset[k]);
		row_bytes[k] = wave64::read_last_lane(inclusive);
		offset[k] = inclusive - offset[k];
	}
	// 3. stage each row's bytes: the 16-byte chunks overlapping [address, address + row_bytes)
	#pragma unroll
	for (unsigned k = 0; k < R; k++) {
		const uintptr_t address = reinterpret_cast<uintptr_t>(compressed_data) + bytes_start;
		const uintptr_t first_chunk = address & ~(uintptr_t) 15;
		const unsigned head = (unsigned) (address - first_chunk);
		const unsigned num_chunks = row_bytes[k] == 0 ? 0 : (head + row_bytes[k] + 15) / 16;   // <= 65
		chunk_t* row_stage = reinterpret_cast<chunk_t*>(stage + k * staging_words);
		if (lane < num_chunks) { row_stage[lane] = memory_access::load_global<chunk_t>(first_chunk + 16 * (uintptr_t) lane); }
		if (num_chunks > 64 && lane == 0) { row_stage[64] = memory_access::load_global<chunk_t>(first_chunk + 16 * (uintptr_t) 64); }
		offset[k] += head;
		bytes_start += row_bytes[k];
	}
	wave64::lds_fence();

	// 4. cut the elements out, 5. store
	#pragma unroll
	for (unsigned k = 0; k < R; k++) {
		const uint64_t i0 = rows_start + (uint64_t) k * row + (uint64_t) lane * E;
		const uint32_t* row_stage = stage + k * staging_words;
		group<U, E> values;
		unsigned at = offset[k];
		#pragma unroll
		for (unsigned j = 0; j < E; j++) {
			const unsigned word = at >> 2, size = sizes[k][j];
			// unconditional aligned reads: the staging area has slack, bytes beyond the element are masked off
			values[j] = (U) cut<UncompressedSize>(row_stage[word], row_stage[word + 1], UncompressedSize == 8 ? row_stage[word + 2] : 0u, at & 3, size, masks);
			at += size;
		}
		if (Full) { store_group(decompressed + i0, values); }
		else if (i0 + E <= segment_end && is_aligned(decompressed + i0, 16)) { store_group(decompressed + i0, values); }
		else {
			#pragma unroll
			for (unsigned j = 0; j < E; j++) { if (i0 + j < segment_end) { decompressed[i0 + j] = values[j]; } }
		}
	}
	wave64::lds_fence();   // the staging area is rewritten by the next call
	return bytes_start;
}

} // namespace detail

namespace detail {

/// The WIDE path: the byte offsets of up to 64 size WORDS' worth of elements from ONE wave scan.
///
/// A lane takes a whole 32-bit word of size fields (2-bit fields, the documented typical case —
/// discard_zero_bytes_variable.cuh:344-347 — 16 elements), so that a wavefront covers a "super-row" of up to
/// 64 * per elements (all of a 1024-element segment) per step:
///   1. lane l: word w0 + l; its byte count = per * min + the sum of its fields, all fields at once (SWAR: two masks,
///      two shifts, a multiply);  ONE 64-lane DPP prefix sum of those (the row-wise path scans once per 64 * E elements);
///   2. the super-row's output rows (64 lanes x E consecutive elements, one 16-byte store per lane, as everywhere): a
///      lane fetches the word and the start of the lane that owns its elements (two ds_bpermute), adds the fields below
///      its own (SWAR again) — and has offsets and sizes without a scan, a division or a size-word load per row;
///   3. the super-row's compressed bytes — ONE contiguous range of at most 4 KiB, known from the scan's total — are staged
///      with at most five coalesced 16-byte-per-lane loads (the row-wise path: one load, with its address arithmetic, per
///      row) while the next super-row's size words are already in flight; elements are cut out of LDS as on the row-wise path.
/// A segment may begin inside a word (field f0 of word w0, f0 a multiple of E): lane 0's fields below f0 are masked out.
/// 64-bit containers of 1-, 2- or 4-bit fields are pairs of such words.  Needs: bits in {1, 2, 3, 4}; (32 / bits) % E == 0;
/// 16-byte aligned rows (the caller checks).
enum : unsigned {
	wide_staging_bytes = 4096,                                   // of a super-row's compressed bytes (rows per step are limited to what fits)
	wide_staging_words = (wide_staging_bytes + 16 + 16) / 4,     // + head misalignment + over-read slack for the aligned dword reads
};

template <unsigned Bits> __device__ __forceinline__ unsigned sum_of_fields(uint32_t x);
template <> __device__ __forceinline__ unsigned sum_of_fields<1>(uint32_t x) { return __builtin_popcount(x); }
template <> __device__ __forceinline__ unsigned sum_of_fields<2>(uint32_t x)
{
	x = (x & 0x33333333u) + ((x >> 2) & 0x33333333u);      // 8 nibbles, each <= 6
	x = (x + (x >> 4)) & 0x0f0f0f0fu;                      // 4 bytes, each <= 12
	return (x * 0x01010101u) >> 24;
}
template <> __device__ __forceinline__ unsigned sum_of_fields<4>(uint32_t x)
{
	x = (x & 0x0f0f0f0fu) + ((x >> 4) & 0x0f0f0f0fu);      // 4 bytes, each <= 30
	return (x * 0x01010101u) >> 24;
}

__host__ __device__ inline bool wide_path_applies(unsigned bits, unsigned container_size, unsigned E)
{
	// (3-bit fields: two steps per 1024-element segment, no faster than the row-wise path; 2-byte elements, E = 8: 6 points slower)
	if (!GIDDY_DZBV_WIDE || bits < 1 || bits > 4 || bits == 3 || E > 4) { return false; }
	return (32 / bits) % E == 0;
}

/// `num_rows` full rows of one segment starting at element `pos`, byte `bytes_start` (both advanced).
template <unsigned Bits, unsigned UncompressedSize>
__device__ __forceinline__ void decompress_wide(
	uint_t<UncompressedSize>* __restrict__ decompressed, const uint8_t* __restrict__ compressed_data,
	const uint32_t* __restrict__ size_words, uint32_t* stage, unsigned lane, uint64_t& pos, unsigned num_rows, uint64_t& bytes_start,
	unsigned min_size, const size_masks& masks)
{
	using namespace memory_access;
	using U = uint_t<UncompressedSize>;
	using chunk_t = typename chunk<16>::type;
	constexpr unsigned E = 16 / UncompressedSize, row = 64 * E, per = 32 / Bits, field_mask = (1u << Bits) - 1;
	constexpr unsigned rows_of_64_words = (64 * per) / row;
	static_assert(per % E == 0 && rows_of_64_words >= 1, "a lane's E size fields share a word");

	// rows per step: what 64 words hold, and what the staging area holds of the largest elements these fields can state
	const unsigned largest = min_size + field_mask < UncompressedSize ? min_size + field_mask : UncompressedSize;
	const unsigned rows_staged = largest == 0 ? rows_of_64_words : wide_staging_bytes / (row * largest);
	const unsigned max_rows = rows_staged < rows_of_64_words ? rows_staged : rows_of_64_words;   // >= 4: a row is at most 1 KiB

	uint64_t word_base = pos / per;
	unsigned field_base = (unsigned) (pos - word_base * per);
	auto rows_of_step = [&](unsigned first_field, unsigned rows_left) {
		const unsigned available = (64 * per - first_field) / row, limit = available < max_rows ? available : max_rows;
		return rows_left < limit ? rows_left : limit;
	};
	auto load_words = [&](uint64_t first_word, unsigned first_field, unsigned rows) {
		const unsigned used = (first_field + rows * row + per - 1) / per;   // words holding those fields (<= 64)
		return lane < used ? size_words[first_word + lane] : 0u;
	};
	unsigned R = rows_of_step(field_base, num_rows);
	uint32_t word = load_words(word_base, field_base, R);

	while (num_rows > 0) {
		// ---- this super-row: R rows = fields [field_base, end_field) of words word_base ...; the next one's words are requested now ----
		const unsigned end_field = field_base + R * row;
		const uint64_t next_word_base = word_base + end_field / per;
		const unsigned next_field_base = end_field % per, next_R = rows_of_step(next_field_base, num_rows - R);
		uint32_t next_word = 0;
		if (next_R > 0) { next_word = load_words(next_word_base, next_field_base, next_R); }

		if (lane == 0) { word &= ~((1u << (field_base * Bits)) - 1u); }   // the segment's first field may be inside the word
		const unsigned lane_bytes = (per - (lane == 0 ? field_base : 0u)) * min_size + sum_of_fields<Bits>(word);
		const unsigned inclusive = wave64::inclusive_sum(lane_bytes);
		// where the lane's word WOULD begin if all of its fields counted: a field's offset = this + field * min + the fields below it
		const unsigned start = inclusive - lane_bytes - (lane == 0 ? field_base * min_size : 0u);
		// the super-row's bytes: up to the offset of field `end_field` (scalar arithmetic on two lanes' values)
		const unsigned last_owner = end_field / per < 63 ? end_field / per : 63, past_field = end_field - last_owner * per;
		const uint32_t last_word = wave64::read_lane(word, last_owner);
		const unsigned super_row_bytes = wave64::read_lane(start, last_owner) + past_field * min_size
			+ sum_of_fields<Bits>(past_field * Bits >= 32 ? last_word : last_word & ((1u << (past_field * Bits)) - 1u));

		// ---- stage the byte range: exactly the 16-byte chunks that overlap it (nothing is read past the buffer's last chunk) ----
		const uintptr_t address = reinterpret_cast<uintptr_t>(compressed_data) + bytes_start;
		const uintptr_t first_chunk = address & ~(uintptr_t) 15;
		const unsigned head = (unsigned) (address - first_chunk);
		const unsigned num_chunks = super_row_bytes == 0 ? 0 : (head + super_row_bytes + 15) / 16;   // <= wide_staging_bytes / 16 + 1
		chunk_t* const staged = reinterpret_cast<chunk_t*>(stage);
		constexpr unsigned pieces = wide_staging_bytes / 1024, ahead = GIDDY_DZBV_WIDE_PIECES_IN_FLIGHT < pieces ? GIDDY_DZBV_WIDE_PIECES_IN_FLIGHT : pieces;
		chunk_t in_flight[ahead];
		const unsigned last_chunk = num_chunks == 0 ? 0 : num_chunks - 1;
		auto request = [&](unsigned piece) {   // (lanes past the end repeat the last chunk)
			if (piece * 64 < num_chunks) {
				const unsigned mine = piece * 64 + lane;
				in_flight[piece % ahead] = load_global<chunk_t>(first_chunk + 16 * (uintptr_t) (mine < last_chunk ? mine : last_chunk));
			}
		};
		#pragma unroll
		for (unsigned piece = 0; piece < ahead; piece++) { request(piece); }   // requests first ...
		#pragma unroll
		for (unsigned piece = 0; piece < pieces; piece++) {                    // ... then into LDS as they arrive
			if (piece * 64 < num_chunks) {
				const unsigned mine = piece * 64 + lane;
				staged[mine < last_chunk ? mine : last_chunk] = in_flight[piece % ahead];
			}
			if (piece + ahead < pieces) { request(piece + ahead); }
		}
		if (num_chunks > pieces * 64 && lane == 0) {   // a full staging area's worth that starts unaligned: one more chunk (rare)
			staged[pieces * 64] = load_global<chunk_t>(first_chunk + 16 * (uintptr_t) (pieces * 64));
		}
		wave64::lds_fence();

		// ---- rows: offsets and sizes from the owner's word and start; elements cut out of the staged range ----
		U* out = decompressed + pos + (uint64_t) lane * E;
		for (unsigned k = 0; k < R; k++) {
			const unsigned g = field_base + k * row + lane * E, owner = g / per, field = g - owner * per;
			const uint32_t owners_word = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (owner * 4), (int) word);
			const unsigned owners_start = (unsigned) __builtin_amdgcn_ds_bpermute((int) (owner * 4), (int) start);
			const unsigned shift = field * Bits;
			unsigned at = head + owners_start + field * min_size + sum_of_fields<Bits>(owners_word & ((1u << shift) - 1u));
			const uint32_t fields = owners_word >> shift;
			group<U, E> values;
			#pragma unroll
			for (unsigned j = 0; j < E; j++) {
				unsigned size = min_size + ((fields >> (j * Bits)) & field_mask);
				if (size > UncompressedSize) { size = UncompressedSize; }   // precondition n_i <= sizeof(element); never read more
				const uint32_t* words = stage + (at >> 2);
				values[j] = (U) cut<UncompressedSize>(words[0], words[1], UncompressedSize == 8 ? words[2] : 0u, at & 3, size, masks);
				at += size;
			}
			store_group(out, values);
			out += row;
		}
		wave64::lds_fence();   // the staging area is rewritten by the next step

		// ---- on to the next super-row ----
		pos += (uint64_t) R * row;
		num_rows -= R;
		bytes_start += super_row_bytes;
		word_base = next_word_base; field_base = next_field_base; R = next_R; word = next_word;
	}
}

} // namespace detail

namespace detail {

/// The fast path: `num_groups` groups of R full rows starting at element `rows_start` (advanced), software-pipelined
/// two deep so that a wavefront never waits for two HBM round trips in a row (CDNA4 retires vector-memory operations
/// in order — loads queue behind earlier stores): while group g is cut out of LDS and stored, group g+1's bytes and
/// group g+2's size words are already in flight.  With SpansSegments the rows may belong to several short segments.
template <bool SpansSegments, unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
__device__ __forceinline__ void decompress_full_groups(
	uint_t<UncompressedSize>* __restrict__ decompressed, const uint8_t* __restrict__ compressed_data,
	const uint_t<ElementSizesContainerSize>* __restrict__ packed_element_sizes,
	uint32_t* stage, unsigned lane, uint64_t& rows_start, uint64_t num_groups, segment_cursor<IndexSize>& cursor, uint32_t rows_per_segment,
	unsigned min_size, unsigned bits, unsigned per, unsigned per_reciprocal, const size_masks& masks)
{
	using container_type = uint_t<ElementSizesContainerSize>;
	using group_t = row_group<UncompressedSize, ElementSizesContainerSize>;
	constexpr unsigned R = rows_in_flight;
	constexpr unsigned group_elements = R * 64 * (16 / UncompressedSize);
	if (num_groups == 0) { return; }

	group_t current, next;
	container_type after_next[R];
	// wave-uniform position of the next group's first size field
	uint64_t container_base = divide(rows_start, per);
	unsigned field_base = (unsigned) (rows_start - container_base * per);
	auto fetch_fields = [&](container_type (&fields)[R]) {
		load_size_fields<UncompressedSize, ElementSizesContainerSize>(fields, packed_element_sizes, lane, container_base, field_base, bits, per, per_reciprocal);
		field_base += group_elements;
		container_base += field_base / per;
		field_base %= per;
	};
	fetch_fields(current.fields);
	if (num_groups > 1) { fetch_fields(next.fields); }
	current.template locate<SpansSegments, IndexSize>(compressed_data, rows_per_segment, cursor, min_size, bits);
	current.request(lane, packed_element_sizes);
	current.stage_into(stage, lane);
	for (uint64_t g = 0; g < num_groups; g++) {
		const bool has_next = g + 1 < num_groups;
		if (has_next) {
			next.template locate<SpansSegments, IndexSize>(compressed_data, rows_per_segment, cursor, min_size, bits);
			next.request(lane, packed_element_sizes);
		}
		if (g + 2 < num_groups) { fetch_fields(after_next); }
		current.extract_and_store(decompressed, stage, lane, rows_start, min_size, bits, masks);
		rows_start += group_elements;
		if (has_next) {
			next.stage_into(stage, lane);
			current = next;
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { next.fields[k] = after_next[k]; }
		}
	}
}

} // namespace detail

template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
// (scalar registers are capped so that eight wavefronts fit a SIMD: measured 3-5 % over what the allocator takes unasked)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(GIDDY_DZBV_SGPRS))) GIDDY_DZBV_OCCUPANCY decompress(
	uint_t<UncompressedSize>*  __restrict__  decompressed,
	const uint8_t*             __restrict__  compressed_data,
	const uint_t<ElementSizesContainerSize>*
	                           __restrict__  packed_element_sizes,
	const uint_t<IndexSize>*   __restrict__  position_anchors,
	size_type_by_index_size<IndexSize>       segment_length,
	size_type_by_index_size<IndexSize>       length,
	element_size_t                           min_represented_element_size,
	native_word_t                            bits_per_element_size)
{
	static_assert(ElementSizesContainerSize == 4 || ElementSizesContainerSize == 8,
		"Only container types with 32 or 64 bits are currently supported");
	constexpr unsigned E = 16 / UncompressedSize;
	constexpr unsigned row = 64 * E;
	constexpr unsigned R = detail::rows_in_flight;
	using container_type = uint_t<ElementSizesContainerSize>;
	__shared__ __attribute__((aligned(16))) uint32_t staging[detail::max_waves_per_block][GIDDY_DZBV_WIDE && detail::wide_staging_words > R * detail::staging_words ? (unsigned) detail::wide_staging_words : R * detail::staging_words];

	__shared__ detail::size_masks masks;
	detail::fill(masks);
	// staging exists for max_waves_per_block wavefronts; in a larger workgroup the extra wavefronts sit idle
	const unsigned working_waves = wave64::per_block() < detail::max_waves_per_block ? wave64::per_block() : detail::max_waves_per_block;
	if (wave64::index_in_block() >= working_waves) { return; }
	uint32_t* stage = staging[wave64::index_in_block()];

	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) blockIdx.x * working_waves + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) gridDim.x * working_waves;
	const uint64_t L = segment_length, n = length;
	if (L == 0 || bits_per_element_size == 0) { return; }
	const unsigned bits = bits_per_element_size;
	const unsigned per = (8 * ElementSizesContainerSize) / bits;
	const unsigned per_reciprocal = per == 1 ? 0u : 0xffffffffu / per + 1;   // ceil(2^32 / per)
	const bool packed = per % E == 0;   // then E consecutive fields starting at a multiple of E share a container
	const bool aligned_output = memory_access::is_aligned(decompressed, 16) && (L % E == 0);

	const uint64_t segments_per_chunk = detail::segments_per_wave_chunk(L, row);
	const uint64_t chunk_length = segments_per_chunk * L;
	const uint64_t group_elements = (uint64_t) R * row;

	// (segments that are not whole rows are taken one per trip: their full groups of rows go the fast way, the tail the slow way)
	const bool fast_path = packed && aligned_output;
	const bool wide_path = fast_path && detail::wide_path_applies(bits, ElementSizesContainerSize, E) && segments_per_chunk == 1 && L < ((uint64_t) 1 << 31) * row;

	for (uint64_t c = wave; ; c += num_waves) {
		const uint64_t chunk_start = c * chunk_length;
		if (chunk_start >= n) { break; }
		const uint64_t chunk_end = (chunk_start + chunk_length < n) ? chunk_start + chunk_length : n;
		const uint64_t num_groups = fast_path ? (chunk_end - chunk_start) / group_elements : 0;
		uint64_t rows_start = chunk_start;
		detail::segment_cursor<IndexSize> cursor;

		if (segments_per_chunk == 1) {
			// one anchored segment: full groups of rows on the fast path, then whatever is left — ragged tails,
			// unaligned outputs, segment lengths that are not whole rows, size fields straddling containers
			cursor.bytes_start = wave64::read_first_lane((uint64_t) position_anchors[c]);
			if constexpr (E <= 4) if (wide_path) {
				const uint32_t* size_words = reinterpret_cast<const uint32_t*>(packed_element_sizes);
				const unsigned full_rows = (unsigned) ((chunk_end - chunk_start) / row);
				auto wide = [&](auto field_bits) {
					detail::decompress_wide<decltype(field_bits)::value, UncompressedSize>(decompressed, compressed_data, size_words, stage, lane,
						rows_start, full_rows, cursor.bytes_start, min_represented_element_size, masks);
				};
				if (bits == 2) { wide(std::integral_constant<unsigned, 2>()); }
				else if (bits == 4) { wide(std::integral_constant<unsigned, 4>()); }
				else if (bits == 1) { wide(std::integral_constant<unsigned, 1>()); }
			}
			if (!wide_path)
			detail::decompress_full_groups<false, IndexSize, UncompressedSize, ElementSizesContainerSize>(
				decompressed, compressed_data, packed_element_sizes, stage, lane, rows_start, num_groups, cursor, 0,
				min_represented_element_size, bits, per, per_reciprocal, masks);
			for (; rows_start < chunk_end; rows_start += group_elements) {
				cursor.bytes_start = detail::decompress_rows<false, false, IndexSize, UncompressedSize, ElementSizesContainerSize>(
					decompressed, compressed_data, packed_element_sizes, stage, lane, rows_start, chunk_end, cursor.bytes_start,
					min_represented_element_size, bits, per, masks);
			}
		}
		else {
			// a batch of short segments, each a whole number of rows: the groups of rows span segments
			const uint32_t rows_per_segment = (uint32_t) (L / row);
			cursor.start_at(position_anchors, c * segments_per_chunk, L, n, (unsigned) segments_per_chunk, lane);
			detail::decompress_full_groups<true, IndexSize, UncompressedSize, ElementSizesContainerSize>(
				decompressed, compressed_data, packed_element_sizes, stage, lane, rows_start, num_groups, cursor, rows_per_segment,
				min_represented_element_size, bits, per, per_reciprocal, masks);
			while (rows_start < chunk_end) {
				if (cursor.rows_left_in_segment == 0) { cursor.begin_segment(rows_per_segment); }
				const uint64_t segment_end_ = rows_start + (uint64_t) cursor.rows_left_in_segment * row;
				const uint64_t segment_end = segment_end_ < chunk_end ? segment_end_ : chunk_end;
				cursor.rows_left_in_segment = 0;
				for (; rows_start < segment_end; rows_start += group_elements) {
					cursor.bytes_start = detail::decompress_rows<false, false, IndexSize, UncompressedSize, ElementSizesContainerSize>(
						decompressed, compressed_data, packed_element_sizes, stage, lane, rows_start, segment_end, cursor.bytes_start,
						min_represented_element_size, bits, per, masks);
				}
				rows_start = segment_end;
			}
		}
	}
}

template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;

	launch_config_resolution_params_t(
		device::properties_t device_properties_, device_function::attributes_t kernel_function_attributes_,
		size_t length_in_elements, size_t segment_length)
	: parent(device_properties_, kernel_function_attributes_)
	{
		if (segment_length == 0) { throw std::invalid_argument("The segment length must be a positive number"); }
		grid_construction_resolution  = warp;              // one wavefront per anchored segment
		length                        = util::div_rounding_up(util::div_rounding_up(length_in_elements, segment_length),
		                                    (size_t) detail::segments_per_wave_chunk(segment_length, 64 * (16 / UncompressedSize)));
		serialization_option          = auto_maximized;    // (batches of short) segments at grid stride
		quanta.threads_in_block       = warp_size;
		upper_limits.threads_in_block = 256;
		keep_gpu_busy_factor          = 1024;  // in practice: one tile per workgroup — the dispatcher hands tiles out in address order, which HBM likes best
	}

	launch_config_resolution_params_t(device::properties_t device_properties_, size_t length_in_elements, size_t segment_length)
	: launch_config_resolution_params_t(device_properties_,
		device_function_t(decompress<IndexSize, UncompressedSize, ElementSizesContainerSize>).attributes(),
		length_in_elements, segment_length) { }
};

} // namespace variable_width
} // namespace discard_zero_bytes
} // namespace decompression
} // namespace kernels
} // namespace cuda
