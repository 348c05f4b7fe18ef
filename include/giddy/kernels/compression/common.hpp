// Device-side COMPRESSORS (an extension: the reference ships decompressors only — README.md:128-130, "Issue #3" — so that
// 2^30-element columns can be put into the formats its kernels read without a host pass).
// The formats are those of libgiddy_amd/formats.py (which cites the reference's kernel documentation per scheme);
// parity for these kernels = that writer's output, byte for byte (tests/test_gpu_compressors.py).
//
// What the variable-rate schemes share: an output position that depends on everything before it.  Three launches on
// the caller's stream, no host synchronisation:
//   1. count    one workgroup per TILE of the column (tiles at grid stride): tile_totals[t] = what tile t will emit
//   2. offsets  ONE workgroup turns the totals into exclusive prefix sums in place; entry [num_tiles] = the grand total
//               (the caller reads the compressed size from there, in its own time)
//   3. write    a workgroup per tile again: recomputes what it emits — the column is read twice, nothing per-element is
//               ever stored in between — prefix-sums it inside the tile (registers + one 64-lane DPP scan + a 4-entry LDS
//               table) and writes at tile_offset + in-tile offset.
// A thread owns `per_thread` CONSECUTIVE elements, so what a tile emits is ordered by thread.  Scratch: the caller's
// `tile_totals`, num_tiles + 1 uint64 entries.
#pragma once
#include <hip/hip_runtime.h>

#include "giddy/common.hpp"
#include "giddy/on_device/wave64.hpp"
#include "giddy/util/math.hpp"

namespace cuda {
namespace kernels {
namespace compression {

enum : unsigned { threads_per_block = 256, waves_per_block = threads_per_block / 64, elements_per_thread = 16, tile_elements = threads_per_block * elements_per_thread };

/// A thread's own elements_per_thread consecutive elements [first, first + elements_per_thread), beyond `n` the `filler`: as 16-byte loads where the
/// thread's range is whole and aligned (element-sized loads at a stride of 16 elements cost the run finders 2-3 x their time)
template <typename T>
__device__ __forceinline__ void load_own_elements(T (&values)[elements_per_thread], const T* __restrict__ input, uint64_t first, uint64_t n, T filler)
{
	constexpr unsigned E = 16 / sizeof(T);   // elements per 16-byte load
	if (first + elements_per_thread <= n && memory_access::is_aligned(input + first, 16)) {
		#pragma unroll
		for (unsigned g = 0; g < elements_per_thread / E; g++) {
			const memory_access::group<T, E> loaded = memory_access::load_group<T, E>(input + first + g * E);
			#pragma unroll
			for (unsigned j = 0; j < E; j++) { values[g * E + j] = loaded[j]; }
		}
		return;
	}
	#pragma unroll
	for (unsigned j = 0; j < elements_per_thread; j++) { values[j] = first + j < n ? input[first + j] : filler; }
}

inline launch_configuration_t tile_launch(uint64_t num_tiles)
{
	const uint64_t cap = 1u << 20;
	return launch_configuration_t((grid_dimension_t) (num_tiles < 1 ? 1 : (num_tiles < cap ? num_tiles : cap)), threads_per_block);
}

/// exclusive prefix of `mine` over the workgroup's threads (thread order) and the workgroup's total; `table`: waves_per_block + 1
/// entries of LDS.  Every thread must call it (two barriers).
template <typename T>
__device__ __forceinline__ T block_exclusive_sum(T mine, T* table, T& total)
{
	const T inclusive = wave64::inclusive_sum(mine);
	__syncthreads();   // (the table may still be read from an earlier call)
	if (wave64::lane_id() == 63) { table[wave64::index_in_block()] = inclusive; }
	__syncthreads();
	T before = 0, all = 0;
	#pragma unroll
	for (unsigned w = 0; w < waves_per_block; w++) {
		const T t = table[w];
		if (w < wave64::index_in_block()) { before += t; }
		all += t;
	}
	total = all;
	return before + inclusive - mine;
}

/// workgroup-wide sum into every thread
template <typename T>
__device__ __forceinline__ T block_sum(T mine, T* table)
{
	T total;
	(void) block_exclusive_sum(mine, table, total);
	return total;
}

/// Phase 2: totals[0 .. count) -> exclusive prefix sums in place, totals[count] = the grand total (also stored to *grand_total
/// when that is not null).  ONE workgroup.
template <typename Total = uint64_t>
__global__ void __launch_bounds__(threads_per_block) exclusive_offsets(Total* __restrict__ totals, uint64_t count, Total* __restrict__ grand_total)
{
	constexpr unsigned per_thread = 16;   // (a sweep covers 4096 entries: 65 536 tiles take 16 sweeps, not 256 — 183 -> 15 us)
	__shared__ Total table[waves_per_block + 1];
	Total carry = 0;
	for (uint64_t base = 0; base < count; base += (uint64_t) threads_per_block * per_thread) {   // (uniform trip count: the barriers inside are safe)
		const uint64_t first = base + (uint64_t) threadIdx.x * per_thread;
		Total mine[per_thread], sum = 0;
		#pragma unroll
		for (unsigned j = 0; j < per_thread; j++) { mine[j] = first + j < count ? totals[first + j] : 0; sum += mine[j]; }
		Total sweep_total;
		Total running = carry + block_exclusive_sum(sum, table, sweep_total);
		#pragma unroll
		for (unsigned j = 0; j < per_thread; j++) {
			if (first + j < count) { totals[first + j] = running; }
			running += mine[j];
		}
		carry += sweep_total;
	}
	if (threadIdx.x == 0) {
		totals[count] = carry;
		if (grand_total != nullptr) { *grand_total = carry; }
	}
}

/// zeroes for the chained kernels' tile states (compression/chained.hpp), in stream order
inline void enqueue_zero(stream::id_t stream, void* memory, size_t bytes)
{
	const hipError_t status = hipMemsetAsync(memory, 0, bytes, stream);
	if (status != hipSuccess) { throw cuda::runtime_error(status, "zeroing a compressor's tile states"); }
}

inline void enqueue_exclusive_offsets(stream::id_t stream, uint64_t* totals, uint64_t count, uint64_t* grand_total = nullptr)
{
	cuda::enqueue_launch(reinterpret_cast<const void*>(exclusive_offsets<uint64_t>), stream, launch_configuration_t(1, threads_per_block), totals, count, grand_total);
}

} // namespace compression
} // namespace kernels
} // namespace cuda
