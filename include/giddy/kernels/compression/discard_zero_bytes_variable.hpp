// Device-side compressor for variable-width discard-zero-bytes (null suppression): element i keeps its n_i = max(min size,
// significant bytes) low-order bytes; n_i - min size goes into a `bits`-wide field, floor(container bits / bits) fields per
// container, LSB first; position_anchors[s] = byte offset of element s * segment_length.
// Format: libgiddy_amd/formats.py::discard_zero_bytes_variable (reference: discard_zero_bytes_variable.cuh:23-33, :332-370).
// A THREAD takes one container's worth of consecutive elements (so a container is built in a register and written once);
// a tile = 256 containers.  Three launches (compression/common.hpp): byte counts per tile, offsets, write — the tile's bytes
// gathered in LDS and written out with 16-byte stores.
#pragma once
#include "giddy/kernels/compression/common.hpp"
#include "giddy/kernels/compression/chained.hpp"


namespace cuda {
namespace kernels {
namespace compression {
namespace discard_zero_bytes {
namespace variable_width {

/// bytes element `value` keeps: max(min size, significant bytes) — capped at the largest size a field can state (a value that needs
/// more is NOT representable with these parameters: it is stored truncated and counted by compress())
template <unsigned UncompressedSize>
__device__ __forceinline__ unsigned bytes_kept(uint_t<UncompressedSize> value, unsigned min_size, unsigned largest_size, bool& representable)
{
	const unsigned significant = value == 0 ? 0 : (64 - (unsigned) __builtin_clzll((unsigned long long) value) + 7) / 8;
	representable = significant <= largest_size;
	const unsigned wanted = significant < min_size ? min_size : significant;
	return wanted > largest_size ? largest_size : wanted;
}

__host__ __device__ __forceinline__ unsigned largest_size_of(unsigned min_size, unsigned bits, unsigned element_size)
{
	const uint64_t largest = (uint64_t) min_size + ((bits >= 32 ? (uint64_t) 1 << 32 : (uint64_t) 1 << bits) - 1);
	return (unsigned) (largest < element_size ? largest : element_size);
}

/// tile_totals[t] = bytes the elements of tile t (256 * per consecutive elements) keep
template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
__global__ void __launch_bounds__(threads_per_block) count_bytes(
	uint64_t*                        __restrict__  tile_totals,
	const uint_t<UncompressedSize>*  __restrict__  input,
	size_type_by_index_size<IndexSize>             length,
	native_word_t                                  min_represented_element_size,
	native_word_t                                  bits_per_element_size)
{
	__shared__ uint64_t table[waves_per_block + 1];
	const unsigned per = (8 * ElementSizesContainerSize) / bits_per_element_size;
	const unsigned largest_size = largest_size_of(min_represented_element_size, bits_per_element_size, UncompressedSize);
	const uint64_t n = length, tile = (uint64_t) threads_per_block * per, num_tiles = (n + tile - 1) / tile;
	for (uint64_t t = blockIdx.x; t < num_tiles; t += gridDim.x) {
		// (a tile's total does not care which thread adds which element: consecutive threads take consecutive elements — coalesced
		// loads, 9 x faster than a thread walking its own container's elements)
		const uint64_t first = t * tile + threadIdx.x;
		uint64_t mine = 0;
		bool representable;
		for (unsigned j = 0; j < per; j++) {
			const uint64_t p = first + (uint64_t) j * threads_per_block;
			if (p < n) { mine += bytes_kept<UncompressedSize>(input[p], min_represented_element_size, largest_size, representable); }
		}
		const uint64_t total = block_sum(mine, table);
		if (threadIdx.x == 0) { tile_totals[t] = total; }
	}
}

/// `tile_offsets`: the exclusive prefix sums of count_bytes' totals
template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize>
__global__ void __launch_bounds__(threads_per_block) compress(
	uint8_t*                                  __restrict__  compressed_data,
	uint_t<ElementSizesContainerSize>*        __restrict__  packed_element_sizes,
	uint_t<IndexSize>*                        __restrict__  position_anchors,
	const uint_t<UncompressedSize>*           __restrict__  input,
	const uint64_t*                           __restrict__  tile_offsets,
	size_type_by_index_size<IndexSize>                      segment_length,
	size_type_by_index_size<IndexSize>                      length,
	native_word_t                                           min_represented_element_size,
	native_word_t                                           bits_per_element_size,
	native_word_t*                            __restrict__  unrepresentable)
{
	using container_type = uint_t<ElementSizesContainerSize>;
	using chunk_t = typename memory_access::chunk<16>::type;
	enum : unsigned { staging_bytes = 40960 };
	// a tile's bytes are gathered here, laid out as they lie in global memory modulo 16, and leave as 16-byte stores
	// (element by element they would be one- to eight-byte stores, scattered: ~5 x slower)
	__shared__ __attribute__((aligned(16))) uint8_t staged[staging_bytes];
	__shared__ uint64_t table[waves_per_block + 1];
	const unsigned bits = bits_per_element_size, min_size = min_represented_element_size;
	const unsigned per = (8 * ElementSizesContainerSize) / bits;
	const unsigned largest_size = largest_size_of(min_size, bits, UncompressedSize);
	const uint64_t n = length, L = segment_length, tile = (uint64_t) threads_per_block * per, num_tiles = (n + tile - 1) / tile;
	const bool through_lds = tile * largest_size + 32 <= staging_bytes;   // (else: straight to global memory, byte by byte)
	unsigned bad = 0;
	for (uint64_t t = blockIdx.x; t < num_tiles; t += gridDim.x) {
		const uint64_t first = t * tile + (uint64_t) threadIdx.x * per;
		uint64_t mine = 0;
		bool representable;
		for (unsigned j = 0; j < per; j++) { if (first + j < n) { mine += bytes_kept<UncompressedSize>(input[first + j], min_size, largest_size, representable); } }
		uint64_t tile_total;
		const uint64_t tile_base = tile_offsets[t];
		const uint64_t before = block_exclusive_sum(mine, table, tile_total);
		uint64_t at = tile_base + before;
		const unsigned misalignment = (unsigned) ((reinterpret_cast<uintptr_t>(compressed_data) + tile_base) & 15);
		unsigned at_staged = misalignment + (unsigned) before;
		container_type container = 0;
		uint64_t until_anchor = first < n ? (L - first % L) % L : 0;   // elements until the next segment's first
		for (unsigned j = 0; j < per && first + j < n; j++) {
			const uint_t<UncompressedSize> value = input[first + j];
			const unsigned size = bytes_kept<UncompressedSize>(value, min_size, largest_size, representable);
			if (!representable) { bad++; }
			if (until_anchor == 0) { position_anchors[(first + j) / L] = (uint_t<IndexSize>) at; until_anchor = L; }
			until_anchor--;
			container |= (container_type) (size - min_size) << (j * bits);
			if (through_lds) {
				for (unsigned k = 0; k < size && k < 8; k++) { staged[at_staged + k] = (uint8_t) ((uint64_t) value >> (8 * k)); }
				at_staged += size;
			}
			else {
				for (unsigned k = 0; k < size && k < 8; k++) { compressed_data[at + k] = (uint8_t) ((uint64_t) value >> (8 * k)); }
			}
			at += size;
		}
		if (first < n) { packed_element_sizes[first / per] = container; }
		if (through_lds) {
			__syncthreads();
			uint8_t* const aligned_start = compressed_data + tile_base - misalignment;   // 16-byte aligned
			const unsigned valid_from = misalignment, valid_to = misalignment + (unsigned) tile_total, chunks = (valid_to + 15) / 16;
			for (unsigned c = threadIdx.x; c < chunks; c += blockDim.x) {
				const unsigned lo = 16 * c, hi = lo + 16;
				if (lo >= valid_from && hi <= valid_to) {
					*reinterpret_cast<chunk_t*>(aligned_start + lo) = *reinterpret_cast<const chunk_t*>(staged + lo);
				}
				else {   // the tile's first and last chunk are shared with its neighbours
					for (unsigned b = (lo < valid_from ? valid_from : lo); b < (hi < valid_to ? hi : valid_to); b++) { aligned_start[b] = staged[b]; }
				}
			}
			__syncthreads();   // (the next tile refills the staging area)
		}
	}
	if (unrepresentable != nullptr && bad != 0) { atomicAdd(unrepresentable, bad); }
}

/// compress() for the shapes that matter (round 4): `per` — fields per container — a power of two from 4 to 64, and a wavefront's bytes
/// (64 * per elements of at most `largest` bytes) within wave_staging_bytes.  Same tiles as count_bytes (threads_per_block * per elements),
/// same outputs bit for bit; what differs is who does what:
///   compress(): a THREAD walks its container's `per` consecutive elements — loads at a stride of a container, the sizes computed twice,
///     the bytes put into LDS one by one: 2.1 ms for 2^28 64-bit elements (1.0 TB/s of the 2.9 GB it moves);
///   here: a WAVEFRONT takes 64 * per consecutive elements as `per` ROWS of 64 — coalesced loads, eight rows in flight; a row's byte
///     offsets are one 64-lane DPP scan of the sizes; a container is the OR of its `per` neighbouring lanes' shifted fields (log2 per
///     butterfly steps); an element's bytes go into the wavefront's zeroed LDS as two or three ds_or_b32 of the value shifted to its byte
///     offset (neighbours share dwords: OR, not store); and the wavefront's bytes leave as 16-byte stores, re-aligned on the way out
///     (aligned LDS dwords + v_alignbyte) to wherever the tiles before it end — known from `tile_offsets` and one barrier per tile.
enum : unsigned { wave_staging_bytes = 8192, rows_in_flight = 8 };

__host__ __device__ inline bool compress_by_rows_applies(unsigned per, unsigned largest_size, uint64_t segment_length)
{
	// (segments of at least 64 elements: at most one segment begins in a row of 64)
	return per >= 4 && per <= 64 && (per & (per - 1)) == 0 && 64u * per * largest_size <= wave_staging_bytes && segment_length >= 64;
}

/// words of LDS a wavefront of compress_by_rows stages its bytes in (+ slack for the re-aligning reads past the last byte; a multiple of 4)
__host__ __device__ inline unsigned compress_by_rows_staging_words(unsigned per, unsigned largest_size)
{
	return ((64u * per * largest_size + 3) / 4 + 4 + 8 + 3) / 4 * 4;
}

/// `Chained`: ONE launch for the whole compressor — no count_bytes, no offsets kernel: `tile_offsets` are then the tiles' STATES (zeroed by
/// the caller), a tile's total is published as soon as its wavefronts have added up their sizes — while its bytes are still being gathered in
/// LDS — and the bytes before the tile come from a decoupled look-back (chained.hpp; bounded by `patience`; a workgroup per tile).  The last
/// tile leaves the column's compressed size in *grand_total (all ones if the chain broke).
template <unsigned IndexSize, unsigned UncompressedSize, unsigned ElementSizesContainerSize, bool Chained = false>
__global__ void __launch_bounds__(threads_per_block) compress_by_rows(
	uint8_t*                                  __restrict__  compressed_data,
	uint_t<ElementSizesContainerSize>*        __restrict__  packed_element_sizes,
	uint_t<IndexSize>*                        __restrict__  position_anchors,
	const uint_t<UncompressedSize>*           __restrict__  input,
	uint64_t*                                 __restrict__  tile_offsets,
	size_type_by_index_size<IndexSize>                      segment_length,
	size_type_by_index_size<IndexSize>                      length,
	native_word_t                                           min_represented_element_size,
	native_word_t                                           bits_per_element_size,
	native_word_t*                            __restrict__  unrepresentable,
	uint64_t*                                 __restrict__  grand_total,
	uint32_t                                                patience)
{
	using container_type = uint_t<ElementSizesContainerSize>;
	using T = uint_t<UncompressedSize>;
	typedef uint32_t chunk_t __attribute__((ext_vector_type(4)));
	// (dynamic LDS: waves_per_block * compress_by_rows_staging_words(per, largest size) words — 4 KiB a wavefront for 2-bit fields of sizes up to 4, so
	// that eight workgroups fit a CU; the worst case the kernel accepts is 8 KiB)
	extern __shared__ __attribute__((aligned(16))) uint32_t staging[];
	__shared__ uint32_t wave_totals[waves_per_block];
	__shared__ uint32_t anchor_offsets[waves_per_block][66];   // the segments that begin among the wavefront's elements (at most one per row + one): byte offset within the wavefront
	const unsigned lane = wave64::lane_id(), wave = wave64::index_in_block();
	const unsigned bits = bits_per_element_size, min_size = min_represented_element_size;
	const unsigned per = (8 * ElementSizesContainerSize) / bits;
	const unsigned largest_size = largest_size_of(min_size, bits, UncompressedSize);
	const uint64_t n = length, L = segment_length, wave_elements = 64ull * per, tile = (uint64_t) threads_per_block * per, num_tiles = (n + tile - 1) / tile;
	const unsigned words_in_use = (64u * per * largest_size + 3) / 4 + 4;
	const unsigned log2_per = (unsigned) __builtin_ctz(per);
	uint32_t* const stage = staging + wave * compress_by_rows_staging_words(per, largest_size);
	unsigned bad = 0;
	// Chained: a workgroup takes `tiles_per_link` CONSECUTIVE tiles and is ONE link of the chain: it adds up the sizes of all their elements
	// first (coalesced 16-byte loads; the write pass re-reads them out of the L2 they just passed through), publishes that, and looks
	// back once — a quarter of the links, each four times as long, is what keeps the look-back short (65 536 links of 32 KiB: 1.5 ms;
	// the three launches: 1.4 ms)
	constexpr unsigned tiles_per_link = Chained ? 4 : 1;
	const uint64_t num_links = (num_tiles + tiles_per_link - 1) / tiles_per_link;
	__shared__ uint64_t slot, link_table[waves_per_block + 1];
	for (uint64_t link = blockIdx.x; link < num_links; link += gridDim.x) {
	uint64_t link_total = 0, bytes_before = 0;   // the link's bytes; the bytes before the tile at hand (known from the first tile's copy-out on)
	if (Chained) {
		constexpr unsigned E = 16 / UncompressedSize;
		const uint64_t first = link * tiles_per_link * tile, end = first + tiles_per_link * tile < n ? first + tiles_per_link * tile : n;
		uint64_t mine = 0;
		bool representable;
		for (uint64_t p = first + (uint64_t) threadIdx.x * E; p < end; p += (uint64_t) threads_per_block * E) {
			if (p + E <= end && memory_access::is_aligned(input + p, 16)) {
				const memory_access::group<T, E> loaded = memory_access::load_group<T, E>(input + p);
				#pragma unroll
				for (unsigned j = 0; j < E; j++) { mine += bytes_kept<UncompressedSize>(loaded[j], min_size, largest_size, representable); }
			}
			else { for (unsigned j = 0; j < E && p + j < end; j++) { mine += bytes_kept<UncompressedSize>(input[p + j], min_size, largest_size, representable); } }
		}
		link_total = block_sum(mine, link_table);
		chained::announce(tile_offsets, link, link_total);
	}
	for (uint64_t t = link * tiles_per_link; t < (link + 1) * tiles_per_link && t < num_tiles; t++) {
		const uint64_t wave_first = t * tile + (uint64_t) wave * wave_elements;
		// (the wavefront's staging area: zeroes — bytes arrive by OR)
		for (unsigned w = lane * 4; w < words_in_use; w += 256) { *reinterpret_cast<chunk_t*>(stage + w) = chunk_t { 0, 0, 0, 0 }; }
		wave64::lds_fence();
		// where the wavefront's first element lies in its segment (wave-uniform; kept by addition from here on), and the first segment that
		// begins at or after it: the segments beginning among the wavefront's elements are that one and its successors, in order
		const uint64_t first_segment_here = wave_first < n ? (wave_first + L - 1) / L : 0;
		uint64_t into_segment = wave_first < n ? wave_first % L : 0;
		unsigned anchors_here = 0;   // segments begun so far (wave-uniform)
		uint32_t running = 0;        // bytes of the wavefront's elements so far
		for (unsigned r0 = 0; r0 < per; r0 += rows_in_flight) {
			T values[rows_in_flight];
			#pragma unroll
			for (unsigned k = 0; k < rows_in_flight; k++) {
				const uint64_t p = wave_first + (uint64_t) (r0 + k) * 64 + lane;
				values[k] = (r0 + k < per && p < n) ? input[p] : (T) 0;
			}
			#pragma unroll
			for (unsigned k = 0; k < rows_in_flight; k++) {
				if (r0 + k >= per) { break; }   // (wave-uniform)
				const uint64_t row_first = wave_first + (uint64_t) (r0 + k) * 64, p = row_first + lane;
				const bool have = p < n;
				bool representable = true;
				const unsigned size = have ? bytes_kept<UncompressedSize>(values[k], min_size, largest_size, representable) : 0u;
				if (!representable) { bad++; }
				const uint32_t inclusive = wave64::inclusive_sum((uint32_t) size);
				const uint32_t offset = running + inclusive - size;
				running += wave64::read_last_lane(inclusive);
				// the container: the OR of its `per` lanes' fields — DPP steps up the group (row_shr 1, 2, 4, 8; row_bcast for groups of 32 and 64
				// lanes), no LDS — which leaves the whole container in the group's LAST lane; that lane writes it
				container_type fields = have ? (container_type) (size - min_size) << ((lane & (per - 1)) * bits) : (container_type) 0;
				fields |= wave64::dpp_or_zero<wave64::row_shr_1>(fields);
				fields |= wave64::dpp_or_zero<wave64::row_shr_2>(fields);
				if (per >= 8) { fields |= wave64::dpp_or_zero<wave64::row_shr_4>(fields); }
				if (per >= 16) { fields |= wave64::dpp_or_zero<wave64::row_shr_8>(fields); }
				if (per >= 32) { fields |= wave64::dpp_or_zero<wave64::row_bcast_15, 0xA>(fields); }
				if (per >= 64) { fields |= wave64::dpp_or_zero<wave64::row_bcast_31, 0xC>(fields); }
				if ((lane & (per - 1)) == per - 1 && row_first + (lane - (per - 1)) < n) { packed_element_sizes[p >> log2_per] = fields; }   // (`per` is a power of two: no 64-bit division per row)
				// a segment's first element: its byte offset is the segment's anchor (at most one per row: segments are 64 elements or longer)
				{
					const uint64_t x = into_segment + lane;   // < L + 64
					const uint64_t who = wave64::ballot(have && (x == 0 || x == L));
					if (who != 0) {
						if ((who >> lane) & 1) { anchor_offsets[wave][anchors_here] = offset; }
						anchors_here++;
					}
					into_segment += 64;
					if (into_segment >= L) { into_segment -= L; }
				}
				// the element's bytes, ORed into the staging area at their byte offset
				if (size != 0) {
					const uint64_t kept = size >= 8 ? (uint64_t) values[k] : (uint64_t) values[k] & ((((uint64_t) 1) << (8 * size)) - 1);
					const unsigned shift = (offset & 3) * 8, word = offset >> 2;
					const uint64_t low = kept << shift;
					__hip_atomic_fetch_or(stage + word, (uint32_t) low, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
					if (size + (offset & 3) > 4) { __hip_atomic_fetch_or(stage + word + 1, (uint32_t) (low >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
					if (UncompressedSize == 8 && size + (offset & 3) > 8) { __hip_atomic_fetch_or(stage + word + 2, (uint32_t) (kept >> (64 - shift)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); }
				}
			}
		}
		// where the wavefront's bytes go: behind the tiles before this one and the wavefronts before this one
		__syncthreads();   // (the table may still be read for the tile before)
		if (lane == 0) { wave_totals[wave] = running; }
		__syncthreads();
		uint64_t wave_base = 0, tile_total = 0;
		#pragma unroll
		for (unsigned w = 0; w < waves_per_block; w++) { if (w < wave) { wave_base += wave_totals[w]; } tile_total += wave_totals[w]; }
		if (Chained) {
			if (t == link * tiles_per_link) {   // the link's first tile: its bytes are in LDS — now, and once per link, the look-back
				bytes_before = chained::exclusive_base(tile_offsets, link, num_links, link_total, &slot, grand_total, true, patience);
				if (bytes_before == chained::failed_base) { return; }   // (the chain broke before this link: every thread sees it; *grand_total says so)
			}
			wave_base += bytes_before;
			bytes_before += tile_total;
		}
		else { wave_base += tile_offsets[t]; }
		wave64::lds_fence();
		for (unsigned i = lane; i < anchors_here; i += 64) { position_anchors[first_segment_here + i] = (uint_t<IndexSize>) (wave_base + anchor_offsets[wave][i]); }
		// out: the 16-byte chunks of global memory that the wavefront's bytes [destination, destination + running) overlap
		const uintptr_t destination = reinterpret_cast<uintptr_t>(compressed_data) + wave_base;
		const uintptr_t aligned_start = destination & ~(uintptr_t) 15;
		const unsigned head = (unsigned) (destination - aligned_start), chunks = running == 0 ? 0 : (head + running + 15) / 16;
		const uint8_t* const staged_bytes = reinterpret_cast<const uint8_t*>(stage);
		for (unsigned c = lane; c < chunks; c += 64) {
			const unsigned lo = 16 * c, hi = lo + 16;   // chunk c holds the wavefront's bytes [lo - head, hi - head)
			if (lo >= head && hi <= head + running) {
				const unsigned q = lo - head, shift = q & 3;
				const uint32_t* words = stage + (q >> 2);
				chunk_t out;
				out.x = __builtin_amdgcn_alignbyte(words[1], words[0], shift); out.y = __builtin_amdgcn_alignbyte(words[2], words[1], shift);
				out.z = __builtin_amdgcn_alignbyte(words[3], words[2], shift); out.w = __builtin_amdgcn_alignbyte(words[4], words[3], shift);
				*reinterpret_cast<chunk_t*>(aligned_start + lo) = out;
			}
			else {   // the first and the last chunk are shared with the neighbours: byte by byte
				for (unsigned b = (lo < head ? head : lo); b < (hi < head + running ? hi : head + running); b++) { reinterpret_cast<uint8_t*>(aligned_start)[b] = staged_bytes[b - head]; }
			}
		}
		wave64::lds_fence();   // (the next tile zeroes the staging area)
	}
	}
	if (unrepresentable != nullptr && bad != 0) { atomicAdd(unrepresentable, bad); }
}

} // namespace variable_width
} // namespace discard_zero_bytes
} // namespace compression
} // namespace kernels
} // namespace cuda
