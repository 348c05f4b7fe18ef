// Building a column's DICTIONARY on the device: its distinct values, sorted ascending — what numpy.unique gives
// libgiddy_amd/formats.py::dictionary.  Dictionaries of up to 65 536 entries (1- and 2-byte keys) in a hash set of 2^18 slots, sorted by
// rank; LARGER ones (4- and 8-byte keys; round 5) in a hash set whose size is an argument — a power of two, some four slots per entry the
// caller makes room for — and sorted by a bitonic network over the collected entries (they are distinct, and a dictionary is a fraction of
// its column: the network's n log^2 n compare-exchanges of 2^20 entries are a millisecond).  Three steps on the caller's stream:
//   1. insert       every element into an open-addressing hash set in global memory (2^18 slots, caller-zeroed): a slot is claimed
//                   by compare-and-swap on its state word, filled, then published; an element whose value is already there — all
//                   but at most 65 536 of them — only reads.  A claimed-but-unpublished slot is retried on the NEXT trip of the
//                   loop, never waited for inside one, so lanes of one wavefront cannot wait for each other.
//   2. collect      the published slots' values, compacted into the entries array (their number through an atomic counter)
//   3. sort         by rank: every entry counts the smaller ones (they are distinct) and goes to that position — or, more than 65 536
//                   entries: pad_for_sorting + the bitonic steps below, in place
// `status[0]` = number of distinct values found (also when there are too many: the caller compares it with the key width's limit),
// `status[1]` = entries collected.
#pragma once
#include "giddy/kernels/compression/common.hpp"

namespace cuda {
namespace kernels {
namespace compression {
namespace dictionary {

enum : unsigned { hash_slots = 1u << 18, max_entries = 1u << 16 };
enum : unsigned { slot_empty = 0, slot_claimed = 1, slot_published = 2 };

__device__ __forceinline__ unsigned hash_of(uint64_t v, unsigned slot_mask = hash_slots - 1)
{
	v ^= v >> 33; v *= 0xff51afd7ed558ccdull; v ^= v >> 33; v *= 0xc4ceb9fe1a85ec53ull; v ^= v >> 33;
	return (unsigned) v & slot_mask;
}

/// `slot_states` (hash_slots x unsigned, zeroed by the caller), `slot_values` (hash_slots x uint64), `status` (2 x unsigned, zeroed)
///
/// Round 4: in front of the global table every WORKGROUP keeps a direct-mapped cache in LDS of values it knows to be published there (4096
/// entries): an element whose value is in it is done after one LDS read — where each element used to cost two dependent loads from L2
/// (state, value: 2.2 ms for 2^28 elements, 0.5 TB/s).  A grid of at most `insert_blocks` workgroups walks the column at grid stride, 16
/// bytes per lane per step, so that a cache is filled once per 2^17 elements or so, not once per 1024.
enum : unsigned { cache_entries = 4096, insert_blocks = 4096 };

__device__ __forceinline__ unsigned cache_index_of(uint64_t v)
{
	return (((uint32_t) v ^ (uint32_t) (v >> 32)) * 0x9E3779B1u) >> 20;   // 12 bits
}

template <unsigned IndexSize, unsigned UncompressedSize>
__global__ void __launch_bounds__(threads_per_block) insert_distinct(
	unsigned*                        __restrict__  slot_states,
	uint64_t*                        __restrict__  slot_values,
	unsigned*                        __restrict__  status,
	const uint_t<UncompressedSize>*  __restrict__  input,
	size_type_by_index_size<IndexSize>             length,
	unsigned                                       num_slots,      // a power of two (0: hash_slots — callers of the five-parameter form)
	unsigned                                       most_entries)   // distinct values beyond which the search gives up (0: max_entries)
{
	using T = uint_t<UncompressedSize>;
	constexpr unsigned E = 16 / UncompressedSize;
	const unsigned slot_mask = (num_slots ? num_slots : (unsigned) hash_slots) - 1, give_up_beyond = most_entries ? most_entries : (unsigned) max_entries;
	__shared__ uint64_t known[cache_entries];   // known[i]: a value with cache_index_of == i that IS published in the global table — or, at first, one whose index is not i
	for (unsigned i = threadIdx.x; i < cache_entries; i += blockDim.x) {
		uint64_t elsewhere = 0;
		while (cache_index_of(elsewhere) == i) { elsewhere++; }
		known[i] = elsewhere;
	}
	__syncthreads();
	auto insert = [&](uint64_t value) {
		const unsigned in_cache = cache_index_of(value);
		if (known[in_cache] == value) { return; }
		unsigned slot = hash_of(value, slot_mask);
		bool there = false;
		// the value is in the global table already, most likely: the STATE through an ordinary cached load (a published slot never changes, and a
		// stale "empty" or "claimed" only sends the thread the careful way below), the VALUE past the L1 — the table's memory is not
		// cleared between calls, so a stale value line could hold an earlier call's entry that happens to equal this thread's value
		// while the slot was just published for another one
		if (slot_states[slot] == slot_published && __atomic_load_n(slot_values + slot, __ATOMIC_RELAXED) == value) { there = true; }
		for (uint64_t trips = 0; !there && trips < 64ull * (slot_mask + 1ull); trips++) {   // (bounded: a full table — far too many distinct values — ends the search)
			const unsigned state = __atomic_load_n(slot_states + slot, __ATOMIC_ACQUIRE);
			if (state == slot_published) {
				if (__atomic_load_n(slot_values + slot, __ATOMIC_RELAXED) == value) { there = true; break; }
				slot = (slot + 1) & slot_mask;
			}
			else if (state == slot_empty) {
				unsigned expected = slot_empty;
				if (__atomic_compare_exchange_n(slot_states + slot, &expected, (unsigned) slot_claimed, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) {
					__atomic_store_n(slot_values + slot, value, __ATOMIC_RELAXED);
					__atomic_store_n(slot_states + slot, (unsigned) slot_published, __ATOMIC_RELEASE);
					atomicAdd(status, 1u);
					there = true;
					break;
				}
				// (lost the slot: look at it again on the next trip)
			}
			// (claimed, not yet published: again on the next trip — its owner does not wait for anybody)
			if (__atomic_load_n(status, __ATOMIC_RELAXED) > give_up_beyond) { break; }   // too many distinct values: give up
		}
		if (there) { known[in_cache] = value; }   // (one 8-byte LDS store: whichever lane's value lands, it is a published one)
	};
	const uint64_t n = length, groups = n / E;
	if (memory_access::is_aligned(input, 16)) {
		for (uint64_t g = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (uint64_t) gridDim.x * blockDim.x) {
			const memory_access::group<T, E> loaded = memory_access::load_group<T, E>(input + g * E);
			#pragma unroll
			for (unsigned j = 0; j < E; j++) { insert((uint64_t) loaded[j]); }
		}
		for (uint64_t i = groups * E + (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) { insert((uint64_t) input[i]); }
	}
	else {
		for (uint64_t i = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t) gridDim.x * blockDim.x) { insert((uint64_t) input[i]); }
	}
}

/// entries[k] = the k-th published slot's value (in no particular order); status[1] = their number (at most `capacity` are written)
template <unsigned UncompressedSize>
__global__ void __launch_bounds__(threads_per_block) collect_distinct(
	uint_t<UncompressedSize>*  __restrict__  entries,
	unsigned*                  __restrict__  status,
	const unsigned*            __restrict__  slot_states,
	const uint64_t*            __restrict__  slot_values,
	unsigned                                 capacity,
	unsigned                                 num_slots)   // (0: hash_slots)
{
	const unsigned slots = num_slots ? num_slots : (unsigned) hash_slots;
	for (unsigned slot = blockIdx.x * blockDim.x + threadIdx.x; slot < slots; slot += gridDim.x * blockDim.x) {
		if (slot_states[slot] == slot_published) {
			const unsigned k = atomicAdd(status + 1, 1u);
			if (k < capacity) { entries[k] = (uint_t<UncompressedSize>) slot_values[slot]; }
		}
	}
}

/// sorted[rank of unsorted[i]] = unsorted[i], rank = the number of smaller entries (the entries are distinct, so the ranks are a
/// permutation): count = min(status[1], capacity) entries, read from the device; every thread ranks one entry against all of them,
/// which pass through LDS a tile at a time.  65 536 entries: 4.3 G comparisons, a fraction of a millisecond.
template <unsigned UncompressedSize>
__global__ void __launch_bounds__(threads_per_block) place_by_rank(
	uint_t<UncompressedSize>*        __restrict__  sorted,
	const uint_t<UncompressedSize>*  __restrict__  unsorted,
	const unsigned*                  __restrict__  status,
	unsigned                                       capacity)
{
	using U = uint_t<UncompressedSize>;
	__shared__ U tile[threads_per_block];
	const unsigned count = status[1] < capacity ? status[1] : capacity;
	for (unsigned base = blockIdx.x * blockDim.x; base < count; base += gridDim.x * blockDim.x) {   // (workgroup-uniform bounds: the barriers are safe)
		const unsigned i = base + threadIdx.x;
		const U mine = i < count ? unsorted[i] : (U) 0;
		unsigned rank = 0;
		for (unsigned first = 0; first < count; first += blockDim.x) {
			__syncthreads();
			if (first + threadIdx.x < count) { tile[threadIdx.x] = unsorted[first + threadIdx.x]; }
			__syncthreads();
			const unsigned here = count - first < blockDim.x ? count - first : blockDim.x;
			for (unsigned j = 0; j < here; j++) { rank += tile[j] < mine; }
		}
		if (i < count) { sorted[rank] = mine; }
	}
}

// ---- more than 65 536 entries: a bitonic network over the collected entries, in place ------------------------------------------------------
//
// entries[0 .. count) are the dictionary in no particular order, count = min(status[1], capacity) known to the DEVICE only; the array has `padded`
// slots, a power of two: pad_for_sorting fills the rest with the largest value (duplicates of a real entry at worst: they sort to the end), and
// the network sorts all `padded` ascending — the first `count` are the dictionary.  Step (k, j) compare-exchanges i and i ^ j for every i with
// bit j clear, ascending where bit k of i is clear; steps whose j is below `in_lds` elements' reach run inside one workgroup's LDS, all of a k's
// remaining ones in one launch.

template <unsigned UncompressedSize>
__global__ void __launch_bounds__(threads_per_block) pad_for_sorting(uint_t<UncompressedSize>* __restrict__ entries, const unsigned* __restrict__ status, unsigned capacity, unsigned padded)
{
	const unsigned count = status[1] < capacity ? status[1] : capacity;
	for (unsigned i = count + blockIdx.x * blockDim.x + threadIdx.x; i < padded; i += gridDim.x * blockDim.x) { entries[i] = (uint_t<UncompressedSize>) ~(uint_t<UncompressedSize>) 0; }
}

enum : unsigned { sorted_in_lds = 2048 };   // entries a workgroup of 256 threads sorts in LDS (8 per thread)

template <unsigned UncompressedSize>
__global__ void __launch_bounds__(threads_per_block) bitonic_step(uint_t<UncompressedSize>* __restrict__ entries, unsigned padded, unsigned j, unsigned k)
{
	using U = uint_t<UncompressedSize>;
	for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < padded / 2; t += gridDim.x * blockDim.x) {
		const unsigned i = ((t & ~(j - 1)) << 1) | (t & (j - 1));   // t with a clear bit inserted at j
		const U a = entries[i], b = entries[i | j];
		const bool ascending = (i & k) == 0;
		if ((a > b) == ascending) { entries[i] = b; entries[i | j] = a; }
	}
}

/// all steps (k, j) with j from min(k / 2, sorted_in_lds / 2) down to 1, for `k` alone — or, `from_the_start`, for every k up to sorted_in_lds (the network's
/// first stages, which stay inside blocks of sorted_in_lds entries)
template <unsigned UncompressedSize>
__global__ void __launch_bounds__(threads_per_block) bitonic_steps_in_lds(uint_t<UncompressedSize>* __restrict__ entries, unsigned padded, unsigned k, bool from_the_start)
{
	using U = uint_t<UncompressedSize>;
	__shared__ U tile[sorted_in_lds];
	for (unsigned base = blockIdx.x * sorted_in_lds; base < padded; base += gridDim.x * sorted_in_lds) {   // (workgroup-uniform bounds: the barriers are safe; padded is a multiple of sorted_in_lds or smaller than it)
		const unsigned here = padded - base < sorted_in_lds ? padded - base : (unsigned) sorted_in_lds;
		for (unsigned i = threadIdx.x; i < here; i += blockDim.x) { tile[i] = entries[base + i]; }
		__syncthreads();
		for (unsigned stage = from_the_start ? 2 : k; stage <= (from_the_start ? (here < sorted_in_lds ? here : (unsigned) sorted_in_lds) : k); stage <<= 1) {
			for (unsigned j = (stage / 2 < here / 2 ? stage / 2 : here / 2); j >= 1; j >>= 1) {
				for (unsigned t = threadIdx.x; t < here / 2; t += blockDim.x) {
					const unsigned i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
					const U a = tile[i], b = tile[i | j];
					const bool ascending = ((base + i) & stage) == 0;
					if ((a > b) == ascending) { tile[i] = b; tile[i | j] = a; }
				}
				__syncthreads();
			}
			if (!from_the_start) { break; }
		}
		for (unsigned i = threadIdx.x; i < here; i += blockDim.x) { entries[base + i] = tile[i]; }
		__syncthreads();
	}
}

} // namespace dictionary
} // namespace compression
} // namespace kernels
} // namespace cuda
