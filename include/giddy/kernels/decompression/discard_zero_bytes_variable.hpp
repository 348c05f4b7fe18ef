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
#include "giddy/on_device/batch_pieces.hpp"
#include "giddy/on_device/wave64.hpp"
#include "giddy/util/math.hpp"

namespace cuda {
namespace kernels {
namespace decompression {
namespace discard_zero_bytes {
namespace variable_width {

using element_size_t = native_word_t;

// What this decoder's tuning settled on (MI355X; DESIGN.md section 4, "DZB-variable, round 2", "What a long launch loses"); no longer build switches:
// the wide path wherever it applies, two-byte elements cut out in pairs by a table of byte permutations, a wavefront's chunks interleaved with its
// workgroup's other wavefronts', `nt` stores (write-through: -1...-4 points), one chunk per wavefront (K > 1, software-pipelined: slower at every size).
/// 1 KiB pieces of a super-row's bytes requested before the first is written to LDS (measured: 2 >= 3 > 4, which costs a wavefront of occupancy)
enum : unsigned { wide_pieces_in_flight = 2 };
/// Register budget of the two kernels: room for 6 wavefronts per SIMD (the allocator's own choice — 65-67 vector registers, 7 wavefronts — measured 1-3 points
/// faster than forcing 64 registers, which turns scalar spills into vector work; 5 or 7 per SIMD: 74-77 % against 77-79), 102 scalar registers
#define GIDDY_DZBV_KERNEL_ATTRIBUTES __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(102))) __attribute__((amdgpu_waves_per_eu(6)))

namespace detail {

enum : unsigned {
	max_waves_per_block = 4,          // the launch parameters cap the workgroup at 256 threads
	rows_in_flight = 2,
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

/// TWO-BYTE elements, two at a time (round 4).  A lane's eight elements per row cost the general path eight times (size from the field,
/// two aligned LDS reads, an alignment, a mask from LDS): 2231 instructions per wavefront against 1622 for four-byte elements that are twice
/// the output bytes per row — issue-bound, 59-64 % of the HBM peak (profiles/r02_pmc_below_70.txt, r04_pmc_below_70.txt).  A PAIR of
/// neighbouring elements is at most four bytes of input and exactly four bytes of output, so it is ONE byte permutation (v_perm_b32) of the
/// four input bytes at the pair's offset; which one — and how far the pair advances the offset — depends only on the pair's two size
/// fields: a table of 2^(2 bits) entries, {selector, bytes}, built once per workgroup for the launch's `min_size` and `bits` (<= 4).
struct pair_entry { uint32_t selector, bytes; };
struct size_masks_and_pairs : size_masks { pair_entry pairs[256]; };

__device__ __forceinline__ void fill_pairs(size_masks_and_pairs& table, unsigned min_size, unsigned bits)
{
	if (bits > 4) { return; }
	const unsigned field_mask = (1u << bits) - 1;
	for (unsigned code = threadIdx.x; code < (1u << (2 * bits)); code += blockDim.x) {
		unsigned s0 = min_size + (code & field_mask), s1 = min_size + ((code >> bits) & field_mask);
		s0 = s0 > 2 ? 2 : s0; s1 = s1 > 2 ? 2 : s1;   // (precondition n_i <= sizeof(element); never read more)
		// output bytes 0, 1: the first element (input bytes 0, 1); bytes 2, 3: the second (input bytes s0, s0 + 1); 0x0c: a zero byte
		const uint32_t b0 = s0 >= 1 ? 0x00u : 0x0cu, b1 = s0 == 2 ? 0x01u : 0x0cu, b2 = s1 >= 1 ? s0 : 0x0cu, b3 = s1 == 2 ? s0 + 1 : 0x0cu;
		table.pairs[code] = pair_entry { b0 | (b1 << 8) | (b2 << 16) | (b3 << 24), s0 + s1 };
	}
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
		unsigned segments_per_chunk, unsigned lane, unsigned anchors_per_segment = 1)
	{
		const bool have = lane < segments_per_chunk && (segment + lane) * segment_length < length;
		const uint64_t mine = have ? (uint64_t) position_anchors[(segment + lane) * anchors_per_segment] : 0;
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

/// Segments SHORTER than a row of 64 lanes x E elements (round 4).  A segment used to be a wavefront's unit: one of 64 four-byte elements filled a quarter
/// of the lanes, had its own anchor load, scan and staging (8 % of the HBM peak; 128 elements: 14 %).  But "the data is not actually broken up into
/// segments: one can decompress it without using the anchors at all, with no change of semantics" (the reference's discard_zero_bytes_variable.cuh:366-369):
/// an anchor is a checkpoint of the ONE running sum of element sizes, so every k-th anchor with segments of k x segment_length is the same column.
/// k makes the coarser segments whole "super-rows" of the wide path (64 size words) when that takes at most 2^16 elements, else whole rows, else
/// stays 1.  Kernel and launch parameters agree (coalescing_of).
__host__ __device__ inline unsigned coalescing_of(uint64_t segment_length, unsigned row, unsigned bits)
{
	if (segment_length == 0 || segment_length >= row) { return 1; }
	const unsigned L = (unsigned) segment_length;
	unsigned low_zeros = 0;
	while (((L >> low_zeros) & 1u) == 0) { low_zeros++; }
	// the least k with k * L a multiple of `unit` (a power of two): unit / gcd(L, unit) — or 0 when k * L would exceed 2^16
	const unsigned super_row = (bits == 1 || bits == 2 || bits == 4) ? 64 * (32 / bits) : 0;
	unsigned k = 0;
	if (super_row != 0) { k = (super_row >> low_zeros) == 0 ? 1 : super_row >> low_zeros; if ((uint64_t) k * L > 65536) { k = 0; } }
	if (k == 0) { k = (row >> low_zeros) == 0 ? 1 : row >> low_zeros; if ((uint64_t) k * L > 65536) { k = 0; } }
	return k == 0 ? 1 : k;
}

/// wave-uniform a / b without the ~100-instruction 64-bit division when both fit 32 bits
__device__ __forceinline__ uint64_t divide(uint64_t a, unsigned b)
{
	return (a >> 32) == 0 ? (uint64_t) ((uint32_t) a / b) : a / b;
}

/// wave-uniform a / b, b > 0: 32-bit when both fit
__device__ __forceinline__ uint64_t divide_any(uint64_t a, uint64_t b)
{
	return ((a | b) >> 32) == 0 ? (uint64_t) ((uint32_t) a / (uint32_t) b) : a / b;
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
		segment_cursor<IndexSize>& cursor, unsigned min_size, unsigned bits, const pair_entry* pairs = nullptr)
	{
		const container_type field_mask = bits >= 8 * ElementSizesContainerSize ? ~(container_type) 0 : (((container_type) 1 << bits) - 1);
		#pragma unroll
		for (unsigned k = 0; k < R; k++) {
			if (SpansSegments) { cursor.before_row(rows_per_segment); }
			unsigned lane_bytes = 0;
			if (UncompressedSize == 2 && pairs != nullptr) {
				#pragma unroll
				for (unsigned d = 0; d < E / 2; d++) { lane_bytes += pairs[(unsigned) (fields[k] >> (2 * bits * d)) & ((1u << (2 * bits)) - 1u)].bytes; }
			}
			else {
				#pragma unroll
				for (unsigned j = 0; j < E; j++) { lane_bytes += size_of<UncompressedSize>(fields[k], j, min_size, bits, field_mask); }
			}
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
		unsigned lane, uint64_t rows_start, unsigned min_size, unsigned bits, const size_masks& masks, const pair_entry* pairs = nullptr) const
	{
		using U = uint_t<UncompressedSize>;
		const container_type field_mask = bits >= 8 * ElementSizesContainerSize ? ~(container_type) 0 : (((container_type) 1 << bits) - 1);
		#pragma unroll
		for (unsigned k = 0; k < R; k++) {
			const uint64_t i0 = rows_start + (uint64_t) k * 64 * E + (uint64_t) lane * E;
			const uint32_t* row_stage = stage + k * staging_words;
			memory_access::group<U, E> values;
			unsigned at = offset[k];
			if constexpr (UncompressedSize == 2) {
				if (pairs != nullptr) {   // (wave-uniform) two elements per step: the pair's four input bytes, one permutation (pair_entry above)
					#pragma unroll
					for (unsigned d = 0; d < E / 2; d++) {
						const pair_entry entry = pairs[(unsigned) (fields[k] >> (2 * bits * d)) & ((1u << (2 * bits)) - 1u)];
						const unsigned word = at >> 2;
						const uint32_t four_bytes = __builtin_amdgcn_alignbyte(row_stage[word + 1], row_stage[word], at & 3);
						values.words[d] = __builtin_amdgcn_perm(0u, four_bytes, entry.selector);
						at += entry.bytes;
					}
					memory_access::store_group<memory_access::non_temporal>(decompressed + i0, values);
					continue;
				}
			}
			#pragma unroll
			for (unsigned j = 0; j < E; j++) {
				const unsigned word = at >> 2, size = size_of<UncompressedSize>(fields[k], j, min_size, bits, field_mask);
				// unconditional aligned reads: the staging area has slack, bytes beyond the element are masked off
				values[j] = (U) cut<UncompressedSize>(row_stage[word], row_stage[word + 1], UncompressedSize == 8 ? row_stage[word + 2] : 0u, at & 3, size, masks);
				at += size;
			}
			memory_access::store_group<memory_access::non_temporal>(decompressed + i0, values);
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
	if (bits < 1 || bits > 4 || bits == 3 || E > 4) { return false; }
	return (32 / bits) % E == 0;
}

/// What a wavefront does next on the wide path: `rows` full rows — a "super-row", at most 64 size words' worth — of one segment.
/// All wave-uniform.
struct wide_step {
	uint64_t pos;            // the first element
	unsigned field_base;     // its size field within word pos / per
	unsigned rows;
	unsigned anchor_slot;    // the lane that holds the segment's anchor when the step begins a segment; no_anchor: the bytes follow the previous step's
	enum : unsigned { no_anchor = ~0u };
};

/// The super-rows of a wavefront's segments, in order: segments first, first + stride, ... (`segments_left` of them), each
/// cut into steps of what 64 words and the staging area hold.  A segment's ragged end (fewer elements than a row) is not planned.
/// FullSteps: every step is 64 whole words (the caller checks) — the position just advances, no division, no row counts.
template <unsigned Bits, unsigned UncompressedSize, bool FullSteps>
struct wide_plan {
	enum : unsigned { E = 16 / UncompressedSize, row = 64 * E, per = 32 / Bits, rows_of_64_words = (64 * per) / row };
	uint64_t L, n, segment_start, between_segments, pos;   // pos: where the next step begins
	unsigned segments_left, max_rows, rows_left, next_slot;

	__device__ __forceinline__ void start(uint64_t first_segment, unsigned num_segments, unsigned stride, uint64_t L_, uint64_t n_, unsigned max_rows_)
	{
		L = L_; n = n_; segment_start = first_segment * L; between_segments = (uint64_t) stride * L; pos = segment_start;
		segments_left = num_segments; max_rows = max_rows_; rows_left = 0; next_slot = 0;
	}
	__device__ __forceinline__ bool next(wide_step& s)
	{
		s.anchor_slot = wide_step::no_anchor;
		while (rows_left == 0) {
			if (segments_left == 0) { return false; }
			if (next_slot != 0) { segment_start += between_segments; }
			pos = segment_start;
			if (FullSteps) { rows_left = (unsigned) (L / row); }   // (the caller keeps segment_length / row below 2^32)
			else {
				const uint64_t end = pos + L < n ? pos + L : n;
				rows_left = (unsigned) ((end - pos) / row);
			}
			s.anchor_slot = next_slot++; segments_left--;
		}
		s.pos = pos;
		if (FullSteps) { s.field_base = 0; s.rows = rows_of_64_words; }
		else {
			s.field_base = (unsigned) (pos % per);
			const unsigned available = (64 * per - s.field_base) / row, limit = available < max_rows ? available : max_rows;
			s.rows = rows_left < limit ? rows_left : limit;
		}
		pos += (uint64_t) s.rows * row; rows_left -= s.rows;
		return true;
	}
};

/// The wide path over a wavefront's segments, software-pipelined ACROSS super-rows and segments: while super-row t is cut out of
/// LDS and stored, the bytes of super-row t+1 — requested BEFORE t's stores: CDNA4 returns vector-memory operations in order, a
/// load issued behind stores waits for their acknowledgement — and the size words of super-row t+2 are in flight.  A wavefront
/// that lives for one 1024-element segment pays anchor -> size words -> scan -> bytes -> stores as one dependent chain per 8 KiB
/// of output (round 2: 64 % of its cycles waiting); one that takes several segments pays it once.
///   FullSteps: every step is exactly 64 words from field 0 (segment_length a multiple of 64 * per, whole segments only,
///   and the staging area holds them): the row loop is unrolled with a known number of stores, which is what lets the
///   compiler's wait for the next step's bytes leave this step's stores in flight (with a store loop of unknown trip count
///   its vmcnt bookkeeping falls back to waiting for the stores as well).
/// Returns where it stopped (element and byte) so that the caller can finish a ragged end the slow way.
template <unsigned Bits, unsigned IndexSize, unsigned UncompressedSize, bool FullSteps>
__device__ __forceinline__ void decompress_wide(
	uint_t<UncompressedSize>* __restrict__ decompressed, const uint8_t* __restrict__ compressed_data,
	const uint32_t* __restrict__ size_words, const uint_t<IndexSize>* __restrict__ position_anchors,
	uint32_t* stage, unsigned lane,
	uint64_t first_segment, unsigned num_segments, unsigned segment_stride, uint64_t L, uint64_t n,
	unsigned min_size, const size_masks& masks, uint64_t& pos_out, uint64_t& bytes_out, unsigned anchors_per_segment = 1)
{
	using namespace memory_access;
	using U = uint_t<UncompressedSize>;
	using chunk_t = typename chunk<16>::type;
	constexpr unsigned E = 16 / UncompressedSize, row = 64 * E, per = 32 / Bits, field_mask = (1u << Bits) - 1;
	constexpr unsigned rows_of_64_words = (64 * per) / row;
	constexpr unsigned pieces = wide_staging_bytes / 1024;
	static_assert(per % E == 0 && rows_of_64_words >= 1, "a lane's E size fields share a word");

	// rows per step: what 64 words hold, and what the staging area holds of the largest elements these fields can state
	const unsigned largest = min_size + field_mask < UncompressedSize ? min_size + field_mask : UncompressedSize;
	const unsigned rows_staged = largest == 0 ? rows_of_64_words : wide_staging_bytes / (row * largest);

	wide_plan<Bits, UncompressedSize, FullSteps> plan;
	plan.start(first_segment, num_segments, segment_stride, L, n, rows_staged < rows_of_64_words ? rows_staged : rows_of_64_words);   // (max_rows >= 4: a row is at most 1 KiB)

	// all of the wavefront's anchors with one load (num_segments <= 64)
	const uint64_t my_anchor = lane < num_segments ? (uint64_t) position_anchors[(first_segment + (uint64_t) lane * segment_stride) * anchors_per_segment] : 0;

	auto load_words = [&](const wide_step& s) {
		if (FullSteps) { return size_words[s.pos / per + lane]; }
		const unsigned used = (s.field_base + s.rows * row + per - 1) / per;   // words holding those fields (<= 64)
		return lane < used ? size_words[s.pos / per + lane] : 0u;
	};

	// a step's scan and byte range
	struct located {
		uint32_t word;           // the lane's size word (fields below the step's first masked out on lane 0)
		unsigned start;          // where the lane's word WOULD begin in the staging area if all of its fields counted
		unsigned total;          // bytes of the step (wave-uniform)
		uintptr_t first_chunk;   // the 16-byte chunks overlapping the byte range (wave-uniform)
		unsigned num_chunks;
	};
	uint64_t bytes_start = 0;
	auto locate = [&](const wide_step& s, uint32_t word) {
		located at;
		if (s.anchor_slot != wide_step::no_anchor) { bytes_start = wave64::read_lane(my_anchor, s.anchor_slot); }
		const unsigned field_base = FullSteps ? 0u : s.field_base;
		if (!FullSteps && lane == 0) { word &= ~((1u << (field_base * Bits)) - 1u); }   // the segment's first field may be inside the word
		const unsigned lane_bytes = (per - (lane == 0 ? field_base : 0u)) * min_size + sum_of_fields<Bits>(word);
		const unsigned inclusive = wave64::inclusive_sum(lane_bytes);
		const unsigned start = inclusive - lane_bytes - (lane == 0 ? field_base * min_size : 0u);
		if (FullSteps) { at.total = wave64::read_last_lane(inclusive); }
		else {
			// up to the offset of field `end_field` (scalar arithmetic on two lanes' values)
			const unsigned end_field = field_base + s.rows * row;
			const unsigned last_owner = end_field / per < 63 ? end_field / per : 63, past_field = end_field - last_owner * per;
			const uint32_t last_word = wave64::read_lane(word, last_owner);
			at.total = wave64::read_lane(start, last_owner) + past_field * min_size
				+ sum_of_fields<Bits>(past_field * Bits >= 32 ? last_word : last_word & ((1u << (past_field * Bits)) - 1u));
		}
		const uintptr_t address = reinterpret_cast<uintptr_t>(compressed_data) + bytes_start;
		at.first_chunk = address & ~(uintptr_t) 15;
		const unsigned head = (unsigned) (address - at.first_chunk);
		at.num_chunks = at.total == 0 ? 0 : (head + at.total + 15) / 16;   // <= wide_staging_bytes / 16 + 1
		at.word = word; at.start = start + head;
		bytes_start += at.total;
		return at;
	};
	// the byte range: exactly the 16-byte chunks that overlap it (nothing is read past the buffer's last chunk; lanes past the end repeat the last one)
	auto request = [&](const located& at, chunk_t (&in_flight)[pieces]) {
		const unsigned last_chunk = at.num_chunks == 0 ? 0 : at.num_chunks - 1;
		#pragma unroll
		for (unsigned piece = 0; piece < pieces; piece++) {
			if (piece * 64 < at.num_chunks) {
				const unsigned mine = piece * 64 + lane;
				in_flight[piece] = load_global<chunk_t>(at.first_chunk + 16 * (uintptr_t) (mine < last_chunk ? mine : last_chunk));
			}
		}
	};
	chunk_t* const staged = reinterpret_cast<chunk_t*>(stage);
	auto into_lds = [&](const located& at, const chunk_t (&in_flight)[pieces]) {
		const unsigned last_chunk = at.num_chunks == 0 ? 0 : at.num_chunks - 1;
		#pragma unroll
		for (unsigned piece = 0; piece < pieces; piece++) {
			if (piece * 64 < at.num_chunks) {
				const unsigned mine = piece * 64 + lane;
				staged[mine < last_chunk ? mine : last_chunk] = in_flight[piece];
			}
		}
		if (at.num_chunks > pieces * 64 && lane == 0) {   // a full staging area's worth that starts unaligned: one more chunk (rare)
			staged[pieces * 64] = load_global<chunk_t>(at.first_chunk + 16 * (uintptr_t) (pieces * 64));
		}
		wave64::lds_fence();
	};
	// one output row: offsets and sizes from the owner lane's word and start; elements cut out of the staged range
	auto emit_row = [&](const located& at, unsigned field_base, unsigned k, U* out) {
		const unsigned g = field_base + k * row + lane * E, owner = g / per, field = g - owner * per;
		const uint32_t owners_word = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (owner * 4), (int) at.word);
		const unsigned owners_start = (unsigned) __builtin_amdgcn_ds_bpermute((int) (owner * 4), (int) at.start);
		const unsigned shift = field * Bits;
		unsigned position = owners_start + field * min_size + sum_of_fields<Bits>(owners_word & ((1u << shift) - 1u));
		const uint32_t fields = owners_word >> shift;
		group<U, E> values;
		#pragma unroll
		for (unsigned j = 0; j < E; j++) {
			unsigned size = min_size + ((fields >> (j * Bits)) & field_mask);
			if (size > UncompressedSize) { size = UncompressedSize; }   // precondition n_i <= sizeof(element); never read more
			const uint32_t* words = stage + (position >> 2);
			values[j] = (U) cut<UncompressedSize>(words[0], words[1], UncompressedSize == 8 ? words[2] : 0u, position & 3, size, masks);
			position += size;
		}
		store_group<non_temporal>(out + (uint64_t) k * row, values);
	};

	auto rows_of = [&](const wide_step& s, const located& at) {
		U* out = decompressed + s.pos + (uint64_t) lane * E;
		if (FullSteps) {
			#pragma unroll
			for (unsigned k = 0; k < rows_of_64_words; k++) { emit_row(at, 0, k, out); }
		}
		else {
			for (unsigned k = 0; k < s.rows; k++) { emit_row(at, s.field_base, k, out); }
		}
		wave64::lds_fence();   // the staging area is rewritten by the next step
#if defined(GIDDY_DZBV_STAMP_PROBE)
		// (experiment: when was this step finished, and on which XCD — in place of the step's first element)
		if (lane == 0) { decompressed[s.pos] = (U) ((__builtin_amdgcn_s_memrealtime() << 4) | (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15)); }
#endif
	};

	// Step t's rows are cut out and stored between the REQUEST for step t+1's bytes and their arrival in LDS: in straight-line code the
	// compiler's wait for those bytes is "all but the last <stores of step t>" — it leaves the stores in flight.  (Were the wait at a
	// point that the prologue reaches too, it would be the stricter of the two paths' counts, i.e. wait for the stores.)
	wide_step current, next, after;
	if (!plan.next(current)) { return; }
	uint32_t next_word = 0, after_word = 0;
	located here = locate(current, load_words(current)), there;
	bool have_next = plan.next(next);
	next_word = load_words(have_next ? next : current);
	chunk_t in_flight[pieces];
	request(here, in_flight);
	into_lds(here, in_flight);
	while (have_next) {
		const bool have_after = plan.next(after);
		after_word = load_words(have_after ? after : next);    // size words of step t + 2 (unconditional: a branch here would make every later wait a wait for all loads)
		there = locate(next, next_word);
		request(there, in_flight);                             // bytes of step t + 1: before the stores of step t
		rows_of(current, here);
		into_lds(there, in_flight);
		current = next; here = there; next = after; next_word = after_word; have_next = have_after;
	}
	rows_of(current, here);
	pos_out = current.pos + (uint64_t) current.rows * row;
	bytes_out = bytes_start;
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
	// (two-byte elements, size fields of at most 4 bits: pairs of elements by table — pair_entry)
	const pair_entry* const pairs = (UncompressedSize == 2 && bits <= 4) ? static_cast<const size_masks_and_pairs&>(masks).pairs : nullptr;
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
	current.template locate<SpansSegments, IndexSize>(compressed_data, rows_per_segment, cursor, min_size, bits, pairs);
	current.request(lane, packed_element_sizes);
	current.stage_into(stage, lane);
	for (uint64_t g = 0; g < num_groups; g++) {
		const bool has_next = g + 1 < num_groups;
		if (has_next) {
			next.template locate<SpansSegments, IndexSize>(compressed_data, rows_per_segment, cursor, min_size, bits, pairs);
			next.request(lane, packed_element_sizes);
		}
		if (g + 2 < num_groups) { fetch_fields(after_next); }
		current.extract_and_store(decompressed, stage, lane, rows_start, min_size, bits, masks, pairs);
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

namespace detail {

template <unsigned UncompressedSize>
struct block_storage {
	enum : unsigned { R = rows_in_flight, words_per_wave = wide_staging_words > R * staging_words ? (unsigned) wide_staging_words : R * staging_words };
	uint32_t staging[max_waves_per_block][words_per_wave];
	typename std::conditional<UncompressedSize == 2, size_masks_and_pairs, size_masks>::type masks;   // (two-byte elements: + the table of pairs)
};

/// What workgroup `block_index` of a grid of `blocks_in_grid` decodes.
template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
__device__ __forceinline__ void decompress_as_block(
	uint64_t block_index, uint64_t blocks_in_grid, uint32_t (&staging)[max_waves_per_block][block_storage<UncompressedSize>::words_per_wave], const size_masks& masks,
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
	constexpr unsigned E = 16 / UncompressedSize;
	constexpr unsigned row = 64 * E;
	constexpr unsigned R = detail::rows_in_flight;
	using container_type = uint_t<ElementSizesContainerSize>;
	// staging exists for max_waves_per_block wavefronts; in a larger workgroup the extra wavefronts sit idle
	const unsigned working_waves = wave64::per_block() < detail::max_waves_per_block ? wave64::per_block() : detail::max_waves_per_block;
	if (wave64::index_in_block() >= working_waves) { return; }
	uint32_t* stage = staging[wave64::index_in_block()];

	const unsigned lane = wave64::lane_id();
	const uint64_t num_waves = blocks_in_grid * working_waves;
	const uint64_t n = length;
	if (segment_length == 0 || bits_per_element_size == 0) { return; }
	const unsigned bits = bits_per_element_size;
	// (short segments: every k-th anchor, segments k times as long — coalescing_of)
	const unsigned anchors_per_segment = detail::coalescing_of(segment_length, row, bits);
	const uint64_t L = (uint64_t) segment_length * anchors_per_segment;
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

	// Chunks (an anchored segment, or a batch of short ones) are handed out in SPANS: with K = ceil(chunks / wavefronts) — 1 when the
	// grid covers the column — workgroup b owns chunks [b * w * K, (b + 1) * w * K), w wavefronts per workgroup, and its wavefront i
	// takes b * w * K + i + j * w, j < K (interleaved, so that at every step the workgroup writes w consecutive chunks) or the K
	// consecutive ones from b * w * K + i * K.  Every grid covers the column; a wavefront with several chunks can overlap their
	// memory round trips (the wide path does), which a grid-stride walk — chunk c, then c + all the wavefronts — cannot.
	const uint64_t total_chunks = detail::divide_any(n + chunk_length - 1, chunk_length);
	const uint64_t K = detail::divide_any(total_chunks + num_waves - 1, num_waves);
	const uint64_t block_first = block_index * working_waves * K;
	const uint64_t block_limit = block_first + working_waves * K < total_chunks ? block_first + working_waves * K : total_chunks;
	// (a wavefront's chunks are interleaved with the other wavefronts' of its workgroup; consecutive ones, or a grid stride over all wavefronts, measured slower)
	const unsigned stride = working_waves;
	const uint64_t first_chunk = block_first + (uint64_t) wave64::index_in_block();
	const uint64_t span_limit = block_limit;
	if (first_chunk >= span_limit) { return; }
	const uint64_t my_chunks = detail::divide_any(span_limit - first_chunk + stride - 1, stride);

	if constexpr (E <= 4) if (wide_path) {
		// the wavefront's segments through one pipeline, up to 64 at a time (their anchors arrive with one load) when segments are whole
		// rows; otherwise one at a time, each one's ragged end — fewer elements than a row — the slow way before the next begins
		const uint32_t* size_words = reinterpret_cast<const uint32_t*>(packed_element_sizes);
		auto wide = [&](auto field_bits) {
			constexpr unsigned Bits = decltype(field_bits)::value, per_word = 32 / Bits, field_mask = (1u << Bits) - 1;
			const unsigned largest = min_represented_element_size + field_mask < UncompressedSize ? min_represented_element_size + field_mask : UncompressedSize;
			// (unrolled row loops of up to 8 rows: 16 — 1-bit fields, 8-byte elements — take 88 vector registers, and every path of the kernel pays for them)
			constexpr bool unrolls = (64 * per_word) / row <= 8;
			const bool full_steps = unrolls && L % (64 * per_word) == 0 && (largest == 0 || detail::wide_staging_bytes / (row * largest) >= (64 * per_word) / row);
			const unsigned max_batch = L % row == 0 ? 64u : 1u;
			for (uint64_t done = 0; done < my_chunks; ) {
				const uint64_t segment = first_chunk + done * stride;
				unsigned batch = my_chunks - done < max_batch ? (unsigned) (my_chunks - done) : max_batch;
				uint64_t stopped_at = 0, bytes_at = 0;
				// (the column's last segment may be incomplete: it is the last of some wavefront's, and goes through the general pipeline by itself)
				const bool last_is_partial = (segment + (uint64_t) (batch - 1) * stride + 1) * L > n;
				if (full_steps && !(last_is_partial && batch == 1)) {
					if (last_is_partial) { batch--; }
					if constexpr (unrolls) detail::decompress_wide<Bits, IndexSize, UncompressedSize, true>(decompressed, compressed_data, size_words, position_anchors, stage, lane,
						segment, batch, stride, L, n, min_represented_element_size, masks, stopped_at, bytes_at, anchors_per_segment);
				}
				else {
					detail::decompress_wide<Bits, IndexSize, UncompressedSize, false>(decompressed, compressed_data, size_words, position_anchors, stage, lane,
						segment, batch, stride, L, n, min_represented_element_size, masks, stopped_at, bytes_at, anchors_per_segment);
				}
				done += batch;
				const uint64_t last = segment + (uint64_t) (batch - 1) * stride, last_start = last * L, last_end = last_start + L < n ? last_start + L : n;
				uint64_t tail = last_start + (last_end - last_start) / row * row;
				if (tail < last_end) {
					if (tail == last_start) { bytes_at = wave64::read_first_lane((uint64_t) position_anchors[last * anchors_per_segment]); }   // (not one full row: nothing of it went through the pipeline)
					for (; tail < last_end; tail += group_elements) {
						bytes_at = detail::decompress_rows<false, false, IndexSize, UncompressedSize, ElementSizesContainerSize>(
							decompressed, compressed_data, packed_element_sizes, stage, lane, tail, last_end, bytes_at,
							min_represented_element_size, bits, per, masks);
					}
				}
			}
		};
		if (bits == 2) { wide(std::integral_constant<unsigned, 2>()); }
		else if (bits == 4) { wide(std::integral_constant<unsigned, 4>()); }
		else if (bits == 1) { wide(std::integral_constant<unsigned, 1>()); }
		return;
	}

	for (uint64_t j = 0; j < my_chunks; j++) {
		const uint64_t c = first_chunk + j * stride;
		const uint64_t chunk_start = c * chunk_length;
		const uint64_t chunk_end = (chunk_start + chunk_length < n) ? chunk_start + chunk_length : n;
		const uint64_t num_groups = fast_path ? (chunk_end - chunk_start) / group_elements : 0;
		uint64_t rows_start = chunk_start;
		detail::segment_cursor<IndexSize> cursor;

		if (segments_per_chunk == 1) {
			// one anchored segment: full groups of rows on the fast path, then whatever is left — ragged tails,
			// unaligned outputs, segment lengths that are not whole rows, size fields straddling containers
			cursor.bytes_start = wave64::read_first_lane((uint64_t) position_anchors[c * anchors_per_segment]);
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
			cursor.start_at(position_anchors, c * segments_per_chunk, L, n, (unsigned) segments_per_chunk, lane, anchors_per_segment);
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

} // namespace detail

template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
__global__ void GIDDY_DZBV_KERNEL_ATTRIBUTES decompress(
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
	__shared__ __attribute__((aligned(16))) detail::block_storage<UncompressedSize> storage;
	if constexpr (UncompressedSize == 2) { detail::fill_pairs(storage.masks, min_represented_element_size, bits_per_element_size); }
	detail::fill(storage.masks);   // (ends with the barrier)
	detail::decompress_as_block<IndexSize, UncompressedSize, ElementSizesContainerSize>(blockIdx.x, gridDim.x, storage.staging, storage.masks,
		decompressed, compressed_data, packed_element_sizes, position_anchors, segment_length, length, min_represented_element_size, bits_per_element_size);
}

/// The pieces of a batch launch (giddy/batch.hpp): `decompress` with its parameters out of the calling workgroup's piece's record, as that piece's block
/// (a kernel of its own rather than batch::decode_pieces<>: the register and occupancy attributes above are part of how this decoder was tuned)
template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
__global__ void GIDDY_DZBV_KERNEL_ATTRIBUTES decompress_pieces(
	const batch::piece_t* __restrict__ pieces, const batch::block_t* __restrict__ blocks)
{
	const batch::located_piece mine = batch::piece_of_this_block(pieces, blocks);
	const element_size_t min_represented_element_size = mine.piece.get<element_size_t>(6);
	const native_word_t bits_per_element_size = mine.piece.get<native_word_t>(7);
	__shared__ __attribute__((aligned(16))) detail::block_storage<UncompressedSize> storage;
	if constexpr (UncompressedSize == 2) { detail::fill_pairs(storage.masks, min_represented_element_size, bits_per_element_size); }
	detail::fill(storage.masks);   // (ends with the barrier)
	detail::decompress_as_block<IndexSize, UncompressedSize, ElementSizesContainerSize>(mine.grid.block(), mine.grid.blocks(), storage.staging, storage.masks,
		mine.piece.get<uint_t<UncompressedSize>*>(0), mine.piece.get<const uint8_t*>(1), mine.piece.get<const uint_t<ElementSizesContainerSize>*>(2),
		mine.piece.get<const uint_t<IndexSize>*>(3), mine.piece.get<size_type_by_index_size<IndexSize>>(4), mine.piece.get<size_type_by_index_size<IndexSize>>(5),
		min_represented_element_size, bits_per_element_size);
}

// ---- long segments: finer anchors from the size fields ----------------------------------------------------------------------
//
// A segment is a dependent chain — an element's bytes begin where its predecessor's end — and the decoder is at its best with one
// 1024-element segment per wavefront and nothing carried from one to the next.  The byte offset of every
// 1024th element follows from the SIZE FIELDS alone (1/32 of the column's bytes for 2-bit fields of 64-bit elements): the wrapper
// has them summed per 1024 elements (`fine::totals`), scanned within each segment on top of the segment's anchor (`fine::anchors`),
// and launches the decoder with a segment length of 1024 on those anchors.  Segments of 4096 / 65536 elements: 67 / 55-62 % of the
// HBM peak before, the rate of 1024-element segments minus the pre-pass now.
namespace fine {

enum : unsigned { sub = 1024, subs_per_block = 4096, threads = 256 };

/// Lane s: the bytes of elements [1024 (first + s), 1024 (first + s + 1)), for the 64 such pieces from `first` on — the size words of
/// all 64 read as ONE contiguous stream of 16-byte-per-lane loads (2-bit fields: 16 KiB in 16 loads), each load's four words summed
/// per lane (SWAR), the lanes that share a piece added up by a butterfly, the pieces' sums collected by lane.
template <unsigned Bits>
__device__ __forceinline__ uint32_t bytes_of_64_pieces(const uint32_t* __restrict__ size_words, uint64_t first, uint64_t length, unsigned min_size, unsigned lane)
{
	typedef uint32_t four_words __attribute__((ext_vector_type(4)));
	constexpr unsigned per = 32 / Bits, words_per_piece = sub / per;              // 32 / 64 / 128 words
	constexpr unsigned lanes_per_piece = words_per_piece / 4;                       // 8 / 16 / 32 lanes hold a piece's words
	constexpr unsigned pieces_per_load = 64 / lanes_per_piece, loads = lanes_per_piece;   // 64 pieces in `loads` wave-wide loads
	constexpr unsigned at_a_time = loads < 16 ? loads : 16;
	const uint64_t first_word = first * words_per_piece;
	const uint64_t num_words = (length + per - 1) / per;
	uint32_t mine = 0;
	#pragma unroll
	for (unsigned c0 = 0; c0 < loads; c0 += at_a_time) {
		uint32_t sums[at_a_time];
		#pragma unroll
		for (unsigned c = 0; c < at_a_time; c++) {
			const uint64_t word = first_word + ((uint64_t) (c0 + c) * 64 + lane) * 4;
			four_words w = { 0, 0, 0, 0 };
			if (word + 4 <= num_words) { w = *reinterpret_cast<const four_words*>(size_words + word); }
			else { for (unsigned j = 0; j < 4; j++) { if (word + j < num_words) { w[j] = size_words[word + j]; } } }
			uint32_t sum = 0;
			#pragma unroll
			for (unsigned j = 0; j < 4; j++) {
				const uint64_t element = (word + j) * per;
				uint32_t fields = w[j];
				unsigned valid = per;
				if (element + per > length) { valid = element < length ? (unsigned) (length - element) : 0u; fields &= valid == 0 ? 0u : (valid * Bits >= 32 ? ~0u : (1u << (valid * Bits)) - 1u); }
				// (sizes above the element's width are a breach of the format's precondition — the decoder clamps them; the sum assumes there are none)
				sum += valid * min_size + detail::sum_of_fields<Bits>(fields);
			}
			sums[c] = sum;
		}
		#pragma unroll
		for (unsigned c = 0; c < at_a_time; c++) {
			uint32_t v = sums[c];
			#pragma unroll
			for (unsigned d = 1; d < lanes_per_piece; d <<= 1) { v += (uint32_t) __shfl_xor((int) v, (int) d, 64); }
			// piece (c0 + c) * pieces_per_load + g is in lanes [g * lanes_per_piece, ...): lane s wants piece s
			const uint32_t collected = (uint32_t) __shfl((int) v, (int) ((lane % pieces_per_load) * lanes_per_piece), 64);
			if (lane / pieces_per_load == c0 + c) { mine = collected; }
		}
	}
	return mine;
}

/// coarser[j] = position_anchors[j * stride]: the anchors of segments `stride` times as long (the anchors are checkpoints of one running sum:
/// the reference's discard_zero_bytes_variable.cuh:366-369)
template <unsigned IndexSize>
__global__ void __launch_bounds__(threads) every_kth_anchor(uint_t<IndexSize>* __restrict__ coarser, const uint_t<IndexSize>* __restrict__ position_anchors, uint64_t count, uint64_t stride)
{
	for (uint64_t j = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (uint64_t) gridDim.x * blockDim.x) { coarser[j] = position_anchors[j * stride]; }
}

/// totals[k] = bytes of elements [1024 k, 1024 (k + 1)) — a wavefront per 64 of them
template <unsigned Bits>
__global__ void __launch_bounds__(threads) totals(
	uint64_t* __restrict__ totals_out, const uint32_t* __restrict__ size_words, uint64_t length, unsigned min_size)
{
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) blockIdx.x * (threads / 64) + wave64::index_in_block(), num_waves = (uint64_t) gridDim.x * (threads / 64);
	const uint64_t num_subs = (length + sub - 1) / sub;
	for (uint64_t k0 = wave * 64; k0 < num_subs; k0 += num_waves * 64) {
		const uint32_t mine = bytes_of_64_pieces<Bits>(size_words, k0, length, min_size, lane);
		if (k0 + lane < num_subs) { totals_out[k0 + lane] = mine; }
	}
}

/// Segments of at most 64 x 1024 elements in ONE launch: a wavefront per 64 / (1024s per segment) whole segments — lane s ends up with
/// the bytes of its 1024, one scan across the lanes, restarted at every segment's first lane on that segment's anchor.
template <unsigned Bits, unsigned IndexSize>
__global__ void __launch_bounds__(threads) anchors_of_short_segments(
	uint_t<IndexSize>* __restrict__ anchors_out, const uint32_t* __restrict__ size_words, const uint_t<IndexSize>* __restrict__ position_anchors,
	uint64_t length, uint64_t segment_length, unsigned min_size)
{
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) blockIdx.x * (threads / 64) + wave64::index_in_block(), num_waves = (uint64_t) gridDim.x * (threads / 64);
	const uint64_t num_subs = (length + sub - 1) / sub, num_segments = (length + segment_length - 1) / segment_length;
	const unsigned subs_per_segment = (unsigned) (segment_length / sub);   // <= 64
	const unsigned segments_per_wave = 64 / subs_per_segment, used_lanes = segments_per_wave * subs_per_segment;
	for (uint64_t segment0 = wave * segments_per_wave; segment0 < num_segments; segment0 += num_waves * segments_per_wave) {
		const uint64_t first = segment0 * subs_per_segment;
		const uint32_t mine = bytes_of_64_pieces<Bits>(size_words, first, length, min_size, lane);
		const unsigned segment_in_wave = lane / subs_per_segment, first_lane_of_segment = segment_in_wave * subs_per_segment;
		const bool active = lane < used_lanes && segment0 + segment_in_wave < num_segments && first + lane < num_subs;
		const uint64_t anchor = active ? (uint64_t) position_anchors[segment0 + segment_in_wave] : 0;
		const uint64_t inclusive = wave64::inclusive_sum((uint64_t) mine);
		const uint64_t exclusive = inclusive - mine;
		const uint64_t at_segment_start = (uint64_t) __shfl((long long) exclusive, (int) (first_lane_of_segment < 64 ? first_lane_of_segment : 63), 64);
		if (active) { anchors_out[first + lane] = (uint_t<IndexSize>) (anchor + exclusive - at_segment_start); }
	}
}

/// anchors_out[k] = byte offset of element 1024 k: the anchor of its segment plus the totals of the segment's earlier 1024s.
/// A workgroup per block of 4096 of a segment's 1024s: first the sum of the segment's blocks before its own, then a scan of its own.
template <unsigned IndexSize>
__global__ void __launch_bounds__(threads) anchors(
	uint_t<IndexSize>* __restrict__ anchors_out, const uint64_t* __restrict__ totals_in, const uint_t<IndexSize>* __restrict__ position_anchors,
	uint64_t length, uint64_t segment_length)
{
	__shared__ uint64_t partial[threads / 64 + 1];
	const uint64_t subs_per_segment = segment_length / sub, num_subs = (length + sub - 1) / sub;
	const uint64_t blocks_per_segment = (subs_per_segment + subs_per_block - 1) / subs_per_block;
	const uint64_t num_segments = (length + segment_length - 1) / segment_length;
	const unsigned lane = wave64::lane_id(), w = wave64::index_in_block();
	for (uint64_t b = blockIdx.x; b < num_segments * blocks_per_segment; b += gridDim.x) {
		const uint64_t segment = b / blocks_per_segment, block_in_segment = b % blocks_per_segment;
		const uint64_t segment_first = segment * subs_per_segment;
		const uint64_t first = segment_first + block_in_segment * subs_per_block;
		const uint64_t limit_in_segment = segment_first + subs_per_segment < num_subs ? segment_first + subs_per_segment : num_subs;
		const uint64_t end = first + subs_per_block < limit_in_segment ? first + subs_per_block : limit_in_segment;
		// the segment's 1024s before this block
		uint64_t before = 0;
		for (uint64_t k = segment_first + threadIdx.x; k < first; k += threads) { before += totals_in[k]; }
		for (unsigned d = 32; d > 0; d >>= 1) { before += (uint64_t) __shfl_xor((long long) before, (int) d, 64); }
		if (lane == 0) { partial[w] = before; }
		__syncthreads();
		uint64_t running = (uint64_t) position_anchors[segment];
		for (unsigned v = 0; v < threads / 64; v++) { running += partial[v]; }
		__syncthreads();
		// this block's: each thread a run of consecutive ones, a workgroup scan of the runs' sums
		constexpr unsigned per_thread = subs_per_block / threads;   // 16
		const uint64_t mine_first = first + (uint64_t) threadIdx.x * per_thread;
		uint64_t values[per_thread], sum = 0;
		#pragma unroll
		for (unsigned j = 0; j < per_thread; j++) { values[j] = mine_first + j < end ? totals_in[mine_first + j] : 0; sum += values[j]; }
		const uint64_t inclusive = wave64::inclusive_sum(sum);
		if (lane == 63) { partial[w] = inclusive; }
		__syncthreads();
		uint64_t offset = running + inclusive - sum;
		for (unsigned v = 0; v < w; v++) { offset += partial[v]; }
		#pragma unroll
		for (unsigned j = 0; j < per_thread; j++) { if (mine_first + j < end) { anchors_out[mine_first + j] = (uint_t<IndexSize>) offset; } offset += values[j]; }
		__syncthreads();
	}
}

} // namespace fine

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
		grid_construction_resolution  = warp;              // a wavefront per span of anchored segments
		// (short segments are taken several at a time, as one: detail::coalescing_of — with 2-bit size fields, the reference's documented typical
		// case; other widths only change which multiple is taken, and the kernel covers the column with any grid)
		segment_length               *= detail::coalescing_of(segment_length, 64 * (16 / UncompressedSize), 2);
		const size_t chunks           = util::div_rounding_up(util::div_rounding_up(length_in_elements, segment_length),
		                                    (size_t) detail::segments_per_wave_chunk(segment_length, 64 * (16 / UncompressedSize)));
		length                        = chunks;             // (one chunk per wavefront: K > 1, software-pipelined across chunks, was slower at every size — profiles/r03_dzbv_segments_per_wavefront.txt)
		serialization_option          = auto_maximized;    // spans of (batches of short) segments; the kernel covers the column with any grid
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
