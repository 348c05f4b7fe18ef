// Wave64 device primitives for CDNA4 (gfx950): lane ids, DPP prefix sums, broadcasts, 16-byte
// vector memory access.  These are the building blocks the kernels in giddy/kernels/** use where the
// reference uses 32-lane shuffles and a two-level shared-memory block scan (SURVEY.md §2b:
// primitives/warp.cuh:127-168, primitives/block.cuh:404-464) — re-thought for a 64-lane wavefront:
// a full-wave inclusive sum is 4 row_shr steps + 2 row_bcast steps on the VALU's DPP path, no LDS.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

namespace cuda {
namespace wave64 {

enum : unsigned { size = 64, log_size = 6 };

__device__ __forceinline__ unsigned lane_id()
{
	return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

/// index of this wavefront within its workgroup / number of wavefronts in the workgroup (1-D blocks).
/// readfirstlane tells the compiler the value is wave-uniform, so everything derived from it (segment
/// numbers, loop bounds, base addresses) lives in scalar registers and branches on it are real branches.
__device__ __forceinline__ unsigned index_in_block() { return (unsigned) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> log_size)); }
__device__ __forceinline__ unsigned per_block() { return (blockDim.x + size - 1) >> log_size; }

// DPP controls (LLVM AMDGPU: row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143)
enum : int { row_shr_1 = 0x111, row_shr_2 = 0x112, row_shr_4 = 0x114, row_shr_8 = 0x118, row_bcast_15 = 0x142, row_bcast_31 = 0x143 };

/// lanes with no valid source (shifted in from outside the row, or masked out by RowMask) read 0
template <int Control, int RowMask = 0xf>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v)
{
	return (uint32_t) __builtin_amdgcn_update_dpp(0, (int) v, Control, RowMask, 0xf, false);
}
template <int Control, int RowMask = 0xf>
__device__ __forceinline__ uint64_t dpp_or_zero(uint64_t v)
{
	uint32_t lo = dpp_or_zero<Control, RowMask>((uint32_t) v);
	uint32_t hi = dpp_or_zero<Control, RowMask>((uint32_t) (v >> 32));
	return ((uint64_t) hi << 32) | lo;
}

/// Inclusive prefix sum across the 64 lanes (wrapping arithmetic), T = uint32_t or uint64_t.
template <typename T>
__device__ __forceinline__ T inclusive_sum(T v)
{
	v += dpp_or_zero<row_shr_1>(v);
	v += dpp_or_zero<row_shr_2>(v);
	v += dpp_or_zero<row_shr_4>(v);
	v += dpp_or_zero<row_shr_8>(v);          // each 16-lane row is scanned
	v += dpp_or_zero<row_bcast_15, 0xA>(v);  // rows 1,3 += last lane of rows 0,2
	v += dpp_or_zero<row_bcast_31, 0xC>(v);  // rows 2,3 += lane 31
	return v;
}

/// Inclusive prefix maximum across the 64 lanes (values >= 0 assumed: identity is 0)
__device__ __forceinline__ uint32_t inclusive_max(uint32_t v)
{
	auto mx = [](uint32_t a, uint32_t b) { return a > b ? a : b; };
	v = mx(v, dpp_or_zero<row_shr_1>(v));
	v = mx(v, dpp_or_zero<row_shr_2>(v));
	v = mx(v, dpp_or_zero<row_shr_4>(v));
	v = mx(v, dpp_or_zero<row_shr_8>(v));
	v = mx(v, dpp_or_zero<row_bcast_15, 0xA>(v));
	v = mx(v, dpp_or_zero<row_bcast_31, 0xC>(v));
	return v;
}

/// DPP move of a 1..8-byte value; lanes with no valid source (or masked out by RowMask) get `otherwise`
template <int Control, int RowMask = 0xf, typename T>
__device__ __forceinline__ T dpp_or(T v, T otherwise)
{
	static_assert(sizeof(T) <= 8, "DPP moves handle values of up to 8 bytes");
	if constexpr (sizeof(T) <= 4) {
		uint32_t bits = 0, other_bits = 0;
		__builtin_memcpy(&bits, &v, sizeof(T)); __builtin_memcpy(&other_bits, &otherwise, sizeof(T));
		uint32_t moved = (uint32_t) __builtin_amdgcn_update_dpp((int) other_bits, (int) bits, Control, RowMask, 0xf, false);
		T result; __builtin_memcpy(&result, &moved, sizeof(T));
		return result;
	}
	else {
		uint64_t bits, other_bits;
		__builtin_memcpy(&bits, &v, 8); __builtin_memcpy(&other_bits, &otherwise, 8);
		uint32_t lo = (uint32_t) __builtin_amdgcn_update_dpp((int) (uint32_t) other_bits, (int) (uint32_t) bits, Control, RowMask, 0xf, false);
		uint32_t hi = (uint32_t) __builtin_amdgcn_update_dpp((int) (uint32_t) (other_bits >> 32), (int) (uint32_t) (bits >> 32), Control, RowMask, 0xf, false);
		uint64_t moved = ((uint64_t) hi << 32) | lo;
		T result; __builtin_memcpy(&result, &moved, 8);
		return result;
	}
}

/// Inclusive prefix reduction across the 64 lanes with any associative, commutative operator that has a
/// neutral value (cuda::functors::plus, minimum, maximum, bit_or ...): the same six DPP steps as inclusive_sum
template <typename Op>
__device__ __forceinline__ typename Op::result_type inclusive_scan(typename Op::result_type v)
{
	using T = typename Op::result_type;
	const T neutral = Op::neutral_value();
	Op op;
	v = op(dpp_or<row_shr_1>(v, neutral), v);
	v = op(dpp_or<row_shr_2>(v, neutral), v);
	v = op(dpp_or<row_shr_4>(v, neutral), v);
	v = op(dpp_or<row_shr_8>(v, neutral), v);
	v = op(dpp_or<row_bcast_15, 0xA>(v, neutral), v);
	v = op(dpp_or<row_bcast_31, 0xC>(v, neutral), v);
	return v;
}

template <typename T>
__device__ __forceinline__ T read_lane(T v, unsigned lane)
{
	if constexpr (std::is_floating_point<T>::value) {
		typename std::conditional<sizeof(T) == 4, uint32_t, uint64_t>::type bits;
		__builtin_memcpy(&bits, &v, sizeof(T));
		bits = read_lane(bits, lane);
		T result; __builtin_memcpy(&result, &bits, sizeof(T));
		return result;
	}
	else if constexpr (sizeof(T) <= 4) { return (T) __builtin_amdgcn_readlane((int) v, (int) lane); }
	else {
		uint32_t lo = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) v, (int) lane);
		uint32_t hi = (uint32_t) __builtin_amdgcn_readlane((int) (uint32_t) ((uint64_t) v >> 32), (int) lane);
		return (T) (((uint64_t) hi << 32) | lo);
	}
}
template <typename T> __device__ __forceinline__ T read_last_lane(T v) { return read_lane(v, 63u); }
/// every lane reads lane `from`'s value (a per-lane index: ds_bpermute_b32, 4 bytes at a time)
template <typename T>
__device__ __forceinline__ T read_lane_of_choice(T v, unsigned from)
{
	static_assert(std::is_integral<T>::value, "integers only");
	if constexpr (sizeof(T) <= 4) { return (T) __builtin_amdgcn_ds_bpermute((int) (from << 2), (int) v); }
	else {
		uint32_t lo = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (from << 2), (int) (uint32_t) v);
		uint32_t hi = (uint32_t) __builtin_amdgcn_ds_bpermute((int) (from << 2), (int) (uint32_t) ((uint64_t) v >> 32));
		return (T) (((uint64_t) hi << 32) | lo);
	}
}
template <typename T>
__device__ __forceinline__ T read_first_lane(T v)
{
	if constexpr (sizeof(T) <= 4) { return (T) __builtin_amdgcn_readfirstlane((int) v); }
	else {
		uint32_t lo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) v);
		uint32_t hi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) ((uint64_t) v >> 32));
		return (T) (((uint64_t) hi << 32) | lo);
	}
}

__device__ __forceinline__ uint64_t ballot(bool predicate) { return __builtin_amdgcn_ballot_w64(predicate); }

/// value of lane (l - 1); lane 0 gets `first`
template <typename T>
__device__ __forceinline__ T shift_up_by_one(T v, T first)
{
	T r = __shfl_up(v, 1, 64);
	return lane_id() == 0 ? first : r;
}

/// orders this wavefront's LDS traffic: writes by any lane before, reads by any lane after
__device__ __forceinline__ void lds_fence()
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

} // namespace wave64

// ---- 16-byte-and-narrower vector access --------------------------------------------------------

namespace memory_access {

template <unsigned Bytes> struct chunk;   // a naturally aligned power-of-two byte group
template <> struct chunk<1>  { using type = uint8_t; };
template <> struct chunk<2>  { using type = uint16_t; };
template <> struct chunk<4>  { using type = uint32_t; };
template <> struct chunk<8>  { using type = uint32_t __attribute__((ext_vector_type(2))); };
template <> struct chunk<16> { using type = uint32_t __attribute__((ext_vector_type(4))); };

/// N elements of T moved as one memory instruction when N*sizeof(T) <= 16 (more: 16-byte pieces).
template <typename T, unsigned N, bool Packed = (sizeof(T) < 4 && (N * sizeof(T)) % 4 == 0)>
struct alignas(N * sizeof(T) >= 16 ? 16 : N * sizeof(T)) group {
	T e[N];
	__device__ __forceinline__ T& operator[](unsigned i) { return e[i]; }
	__device__ __forceinline__ const T& operator[](unsigned i) const { return e[i]; }
};

/// 1- and 2-byte elements filling whole dwords are kept AS dwords and cut out / put in with shifts: left as an
/// array of bytes the compiler gives every element a vector register of its own (a lane's sixteen 1-byte
/// elements: 16 registers instead of 4, and the narrow decoders lose half their occupancy or spill).
template <typename T, unsigned N>
struct alignas(N * sizeof(T) >= 16 ? 16 : N * sizeof(T)) group<T, N, true> {
	enum : unsigned { per_word = 4 / sizeof(T), bits = 8 * sizeof(T), mask = (1u << bits) - 1 };
	uint32_t words[N / per_word];

	struct reference {
		uint32_t& word; unsigned shift;
		__device__ __forceinline__ operator T() const { return (T) ((word >> shift) & mask); }
		__device__ __forceinline__ reference& operator=(T v)
		{
			word = (word & ~((uint32_t) mask << shift)) | (((uint32_t) v & mask) << shift);
			return *this;
		}
		__device__ __forceinline__ reference& operator=(const reference& other) { return *this = (T) other; }
	};
	__device__ __forceinline__ reference operator[](unsigned i) { return reference { words[i / per_word], (i % per_word) * bits }; }
	__device__ __forceinline__ T operator[](unsigned i) const { return (T) ((words[i / per_word] >> ((i % per_word) * bits)) & mask); }
};

template <typename T, unsigned N>
__device__ __forceinline__ group<T, N> load_group(const T* __restrict__ p)
{
	constexpr unsigned bytes = N * sizeof(T);
	constexpr unsigned piece = bytes >= 16 ? 16 : bytes;
	using piece_t = typename chunk<piece>::type;
	group<T, N> g;
	#pragma unroll
	for (unsigned i = 0; i < bytes / piece; i++) {
		reinterpret_cast<piece_t*>(&g)[i] = reinterpret_cast<const piece_t*>(p)[i];
	}
	return g;
}

/// (Non-temporal LOADS — round 5, tools/read_probe.hip: a read stream alone gets 83-87 % of the HBM peak with them and 74-78 % with plain loads, and the run compressors' one
/// pass over the column takes 190-198 us with them against 205-225 — were tried for every load_group of the decoders, beside their stores, at 2^30 elements: Dictionary
/// -1.3 ... -2 points, Frame of Reference, the run decoders and both discard-zero-bytes forms +-0.3, Delta +1 over its unpatched instantiations and +-3 — noise — over the
/// fused ones: the decoders keep plain loads.  Nor did the other compressors gain: DZB-variable 1.22-1.26 ms against 1.17-1.19, dictionary encoding and construction the same.)

/// ... from an address aligned for T only (a column handed over at an odd element offset; gfx950 takes wide loads of global memory at any address — at
/// the price of the second line a piece may lie across)
template <typename T, unsigned N>
__device__ __forceinline__ group<T, N> load_group_at_any_alignment(const T* __restrict__ p)
{
	constexpr unsigned bytes = N * sizeof(T);
	constexpr unsigned piece = bytes >= 16 ? 16 : bytes;
	typedef typename chunk<piece>::type loose_piece_t __attribute__((aligned(sizeof(T))));
	using piece_t = typename chunk<piece>::type;
	group<T, N> g;
	#pragma unroll
	for (unsigned i = 0; i < bytes / piece; i++) {
		reinterpret_cast<piece_t*>(&g)[i] = reinterpret_cast<const loose_piece_t*>(p)[i];
	}
	return g;
}

/// How a 16-byte store treats the caches.  `non_temporal` (`nt`): the line stays in the XCD's L2 marked for early eviction — the
/// acknowledgement comes back soonest, which is what a wavefront wants that issues LOADS after its stores (the counter is in-order: the run
/// decoders' look-ups through L1/L2).  `streaming` (`sc1 nt`): write-through, the line is dropped — for output that nobody reads again and
/// wavefronts that only store from then on: measured +2.5 points for Delta, +4 for the staged run decoders, +1.5...2 for Dictionary / FOR,
/// +1.2 for Model, and -5 for the look-up run decoders, -14 for DZB-fixed <4,8,2>, -1 for DZB-variable.
enum store_policy : int { plain_store = 0, non_temporal = 1, streaming = 2 };

/// the policy a decoder family asks for (kernels/tuning.hpp), for outputs of 4 bytes or more only: with 1- and 2-byte outputs a workgroup's
/// tile is a quarter or half the bytes and the slower acknowledgement outweighs the gain (Dictionary <4,1,1>, <4,2,2>: -6 points; Delta <4,2,1>: -3)
template <unsigned OutputSize> constexpr int wide_output_policy(int requested) { return OutputSize >= 4 ? requested : (int) non_temporal; }

template <int Policy = non_temporal, typename T, unsigned N>
__device__ __forceinline__ void store_group(T* __restrict__ p, const group<T, N>& g)
{
	constexpr unsigned bytes = N * sizeof(T);
	constexpr unsigned piece = bytes >= 16 ? 16 : bytes;
	using piece_t = typename chunk<piece>::type;
	#pragma unroll
	for (unsigned i = 0; i < bytes / piece; i++) {
		if constexpr (Policy == streaming && piece == 16) {
			// (no builtin names this cache policy for a flat pointer.  The compiler does not count the store — which only makes its own
			// waits stricter — and pads no hazard inside the string: a store of more than 8 bytes reads its data registers up to
			// two states after issue, hence the s_nop before the compiler's next instruction may overwrite them)
			typedef unsigned int four_words __attribute__((ext_vector_type(4)));
			const four_words data = reinterpret_cast<const four_words*>(&g)[i];
			piece_t* address = reinterpret_cast<piece_t*>(p) + i;
			asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" :: "v"(address), "v"(data) : "memory");
		}
		else if constexpr (Policy != plain_store) { __builtin_nontemporal_store(reinterpret_cast<const piece_t*>(&g)[i], reinterpret_cast<piece_t*>(p) + i); }
		else { reinterpret_cast<piece_t*>(p)[i] = reinterpret_cast<const piece_t*>(&g)[i]; }
	}
}

/// a load from GLOBAL memory at an address held as an integer.  Dereferencing a pointer made from an integer is a
/// FLAT load to the compiler — it counts against the LDS counter as well as the vector-memory one, so every wait for
/// an LDS read becomes a wait for all loads in flight; naming the address space keeps the two apart.
template <typename T>
__device__ __forceinline__ T load_global(uintptr_t address)
{
	return *reinterpret_cast<const T __attribute__((address_space(1)))*>(address);
}

template <typename P>
__host__ __device__ __forceinline__ bool is_aligned(const P* p, unsigned bytes) { return (reinterpret_cast<uintptr_t>(p) & (bytes - 1)) == 0; }

} // namespace memory_access
} // namespace cuda
