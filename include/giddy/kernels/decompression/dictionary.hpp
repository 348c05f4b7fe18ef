// Dictionary decompression for CDNA4:   out[p] = dictionary_entries[compressed_input[p]]
//
// Kernel template name and parameter list as in the reference's
// src/kernels/decompression/dictionary.cuh:61-75 (which forwards to gather,
// src/kernels/data_layout/gather.cuh:16-146).  Execution shape, re-designed:
//
//  * the reference reads one key per thread and chases it with a dependent global load of the
//    entry; its shared-memory dictionary cache is commented out and force-disabled
//    (gather.cuh:33-90, :202-205);
//  * here the dictionary is staged ONCE per workgroup into LDS (the launch configuration asks for
//    num_entries * sizeof(entry) bytes of dynamic LDS whenever that fits); with a small dictionary
//    every workgroup takes one tile, with a large one workgroups walk enough tiles at grid stride to
//    amortise the staging (see the launch parameters below).  A lane loads G keys as a single 4/8/16-byte load, performs G LDS look-ups
//    (ds_read; random keys cost a few bank-conflict cycles, far below the HBM budget of ~10 B/clk/CU)
//    and writes 16-byte stores that are contiguous across the wavefront.
//  * dictionaries that do not fit in LDS fall back to look-ups through L1/L2.
//
// HBM traffic per element: sizeof(key) read + sizeof(entry) written (+ the dictionary once per
// workgroup, L2-resident) — 5 B for 8-bit keys -> 32-bit values.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>

#include "giddy/common.hpp"
#include "giddy/kernels/launch_config_resolution_params.hpp"
#include "giddy/kernels/tuning.hpp"
#include "giddy/on_device/batch_pieces.hpp"
#include "giddy/on_device/stream_tiles.hpp"
#include "giddy/on_device/wave64.hpp"
#include "giddy/util/math.hpp"

#ifndef GIDDY_DICT_RESTAGE_ENTRIES_PER_THREAD
#define GIDDY_DICT_RESTAGE_ENTRIES_PER_THREAD 16  // entries a thread carries from its request to the LDS write (overlapped staging)
#endif
#ifndef GIDDY_DICT_RESTAGE_UP_TO_BYTES
#define GIDDY_DICT_RESTAGE_UP_TO_BYTES 32768     // larger staged dictionaries run on a persistent grid that stages once per workgroup
#endif

namespace cuda {
namespace kernels {
namespace decompression {
namespace dictionary {

/// the reference's (ignored) run-time serialization factor default, gather.cuh:9-11
constexpr serialization_factor_t get_default_serialization_factor(bool cache_all_data_in_shared_memory)
{
	return cache_all_data_in_shared_memory ? 128 : 16;
}

namespace detail {

/// out[p] = lookup(keys[p]) over the shared streaming tile loop
/// whether an instantiation runs by wavefront chunks (stream_tiles::transform_and_patch) rather than by workgroup tiles: the
/// fused ones do.  (Their 1-byte keys of 8-byte values come in with 16-byte loads through LDS instead of 2 bytes per lane and
/// row: 57 -> 81 % of HBM peak with 2 resident workgroups per CU.  The plain kernel in the same shape stays at its tiles' 70-73 %;
/// round 5, the headline column <4,4,1> at 2^30 through the fused kernel with NO patches: 74 % against the tiles' 80 — tools/experiments/dictionary_plain_vs_fused.py.)
template <bool Patched, unsigned UncompressedSize = 0, unsigned DictionaryIndexSize = 0, bool RestagedDictionary = false> constexpr bool by_wave_chunks()
{
	// (plain 8-byte values with 2-byte or wider keys as well where every workgroup re-stages the dictionary: 32 KiB are then staged
	// per 128 KiB of output instead of per 32 KiB tile — <4,8,2> 60-64 -> 75, <4,8,4> 71 -> 78 % of HBM peak; staged once or looked up
	// through L2, tiles are 3 points better; 4-byte values lose 2-5 points by chunks)
	return Patched || (RestagedDictionary && UncompressedSize == 8 && DictionaryIndexSize >= 2);
}
/// ... whose chunks' keys come in through LDS when a row takes at most this many key bytes per lane: 4 for 8-byte values (1- and
/// 2-byte keys: 55-58 -> 81-83 % and 69 -> 80 % of HBM peak, at 2 resident workgroups per CU), 2 otherwise (4-byte values with
/// 1-byte keys lose 6 points staged)
template <unsigned UncompressedSize> constexpr unsigned stage_keys_up_to() { return UncompressedSize == 8 ? 4 : 2; }
template <unsigned UncompressedSize, unsigned DictionaryIndexSize> constexpr bool stages_chunk_keys()
{
	return stream_tiles::stages_chunk_input<uint_t<DictionaryIndexSize>, uint_t<UncompressedSize>, stage_keys_up_to<UncompressedSize>()>();
}
template <unsigned UncompressedSize, unsigned DictionaryIndexSize>
using chunk_key_storage = stream_tiles::chunk_input_storage<uint_t<DictionaryIndexSize>, uint_t<UncompressedSize>, stages_chunk_keys<UncompressedSize, DictionaryIndexSize>()>;

/// 16-byte stores per lane per tile: 4 (stream_tiles::steps_per_tile) — 8 where a dictionary of 2-byte keys is re-staged per workgroup for 2- and 4-byte values: up to 8 KiB /
/// 16 KiB of dictionary staged per 32 / 64 KiB of output instead of per 16 / 32 (round 6, same-box: <*,2,2> 68 -> 70.5 %, <*,4,2> 75 -> 77 % of the HBM peak; every other
/// instantiation loses with 8: <*,8,1> -10 points, <*,4,4> -5, 1-byte keys -1 ... -3).  Kernel and launch parameters agree.
template <unsigned UncompressedSize, unsigned DictionaryIndexSize, bool RestagedDictionary>
constexpr unsigned steps_per_tile_for()
{
	return (RestagedDictionary && DictionaryIndexSize == 2 && UncompressedSize >= 2 && UncompressedSize <= 4) ? 8u : (unsigned) stream_tiles::steps_per_tile;
}

/// (`LooksUpThroughMemory`: the dictionary is not in LDS — a tile's look-ups all go out before its first store)
template <unsigned UncompressedSize, unsigned DictionaryIndexSize, bool RestagedDictionary = false, bool AnyAlignment = false, bool LooksUpThroughMemory = false, typename Lookup, typename Patches, typename AfterLoads = stream_tiles::nothing_to_do, typename Grid = launch_grid::whole>
__device__ __forceinline__ void translate(
	uint_t<UncompressedSize>*           __restrict__  decompressed,
	const uint_t<DictionaryIndexSize>*  __restrict__  compressed_input,
	uint64_t                                          n,
	Lookup                                            lookup,
	Patches                                           patches,
	chunk_key_storage<UncompressedSize, DictionaryIndexSize>* chunk_staging,
	AfterLoads                                        after_loads = AfterLoads(),
	Grid                                              grid = Grid())
{
	using key_type = uint_t<DictionaryIndexSize>;
	using value_type = uint_t<UncompressedSize>;
	constexpr unsigned G = stream_tiles::group_size<value_type>::value;
	auto produce_group = [&](uint64_t, const memory_access::group<key_type, G>& keys, memory_access::group<value_type, G>& out) {
		#pragma unroll
		for (unsigned j = 0; j < G; j++) { out[j] = lookup(keys[j]); }
	};
	auto produce_one = [&](uint64_t, key_type key) { return lookup(key); };
	if constexpr (!AnyAlignment && by_wave_chunks<Patches::enabled, UncompressedSize, DictionaryIndexSize, RestagedDictionary>()) {   // (any alignment: tiles only)
		if constexpr (Patches::enabled) {
			stream_tiles::transform_and_patch<key_type, value_type, stream_tiles::rows_in_flight, stage_keys_up_to<UncompressedSize>()>(decompressed, compressed_input, n, produce_group, produce_one, patches, after_loads, chunk_staging, grid);
		}
		else {
			// (the plain kernels never have room to stage a chunk's keys: said at compile time, so that the pipelined chunk loop holds no registers for that form —
			// with both forms compiled in, plain <*,8,2> fell from 76 to 61 % of the HBM peak)
			stream_tiles::transform_and_patch<key_type, value_type, stream_tiles::rows_in_flight, 0>(decompressed, compressed_input, n, produce_group, produce_one, patches, after_loads, nullptr, grid);
		}
	}
	else {
		stream_tiles::transform<key_type, value_type, steps_per_tile_for<UncompressedSize, DictionaryIndexSize, RestagedDictionary>(), memory_access::wide_output_policy<sizeof(value_type)>(GIDDY_DICT_STORE_POLICY), AnyAlignment, LooksUpThroughMemory>(decompressed, compressed_input, n, produce_group, produce_one, after_loads, stream_tiles::nothing_to_prepare(), grid);
	}
}

#ifndef GIDDY_DICT_WIDE_RESIDENT_BLOCKS_CAP
#define GIDDY_DICT_WIDE_RESIDENT_BLOCKS_CAP 6   // resident workgroups per CU for 4-8-byte values and small dictionaries (0: no cap): with write-through stores 6-7 are 1.5-2.4 points faster than 8
#endif

template <unsigned UncompressedSize, unsigned DictionaryIndexSize, typename Patches, bool AnyAlignment = false, typename Grid = launch_grid::whole>
__device__ __forceinline__ void decompress(
	uint_t<UncompressedSize>*           __restrict__  decompressed,
	const uint_t<DictionaryIndexSize>*  __restrict__  compressed_input,
	const uint_t<UncompressedSize>*     __restrict__  dictionary_entries,
	uint64_t                                          length,
	uint64_t                                          num_dictionary_entries,
	bool                                              cache_dictionary_in_shared_memory,
	Patches                                           patches,
	bool                                              chunk_key_staging_in_dynamic_lds = false,   // (behind the dictionary)
	Grid                                              grid = Grid())                              // (launch_grid.hpp: the launch's own, or a batch piece's part of it)
{
	using value_type = uint_t<UncompressedSize>;
	using key_type = uint_t<DictionaryIndexSize>;
	extern __shared__ __attribute__((aligned(16))) unsigned char dynamic_lds[];
	// (narrow keys of 8-byte values in the wave-chunk shape: a chunk's keys come in wide and go through LDS — stream_tiles.hpp)
	chunk_key_storage<UncompressedSize, DictionaryIndexSize>* chunk_staging = nullptr;
	if constexpr (Patches::enabled && stages_chunk_keys<UncompressedSize, DictionaryIndexSize>()) {
		if constexpr (DictionaryIndexSize == 1) {
			// (a dictionary of at most 256 entries never competes for the LDS: static storage — measured 4 points faster than dynamic)
			__shared__ chunk_key_storage<UncompressedSize, DictionaryIndexSize> chunk_staging_storage;
			chunk_staging = &chunk_staging_storage;
		}
		else {
			// (wider keys: only where the launch reserved dynamic LDS for it behind the dictionary — launches of dictionaries staged
			// once on a persistent grid, or looked up through L2, do not; the wrapper says which)
			if (chunk_key_staging_in_dynamic_lds) {
				const uint64_t dictionary_bytes_in_lds = cache_dictionary_in_shared_memory ? (num_dictionary_entries * UncompressedSize + 15) / 16 * 16 : 0;
				chunk_staging = reinterpret_cast<chunk_key_storage<UncompressedSize, DictionaryIndexSize>*>(dynamic_lds + dictionary_bytes_in_lds);
			}
		}
	}

	// (1-byte keys: at most 256 entries, one per thread of the usual workgroup — no need to hold registers for more)
	constexpr unsigned entries_per_thread = DictionaryIndexSize == 1 ? 1 : GIDDY_DICT_RESTAGE_ENTRIES_PER_THREAD;
	if (cache_dictionary_in_shared_memory && num_dictionary_entries <= (uint64_t) blockDim.x * entries_per_thread) {
		// A dictionary of a few entries per thread: the entries are REQUESTED now, but put into LDS (and the barrier
		// taken) only after the first tile's key loads have been issued — the two memory round trips overlap
		// instead of following each other.
		value_type* staged = reinterpret_cast<value_type*>(dynamic_lds);
		value_type my_entries[entries_per_thread];
		#pragma unroll
		for (unsigned i = 0; i < entries_per_thread; i++) {
			const uint64_t at = (uint64_t) i * blockDim.x + threadIdx.x;
			if (at < num_dictionary_entries) { my_entries[i] = dictionary_entries[at]; }
		}
		// (stream_tiles calls the hook once per thread, from one place outside its loops — behind the loads of the workgroup's first tile /
		// the wavefront's first chunk: one textual __syncthreads() that every thread of the workgroup reaches on the same path.  If the
		// workgroup is smaller than the launch parameters assumed — a caller's block-size limit — the test above fails and the plain
		// staging loop below runs instead: correct, with several tiles per workgroup.)
		auto stage = [&]() {
			#pragma unroll
			for (unsigned i = 0; i < entries_per_thread; i++) {
				const uint64_t at = (uint64_t) i * blockDim.x + threadIdx.x;
				if (at < num_dictionary_entries) { staged[at] = my_entries[i]; }
			}
			__syncthreads();
		};
		if constexpr (Patches::enabled) {
			// The fused kernels stage BEFORE their chunk loop, with nothing overlapped: a wavefront's chunk begins with the seek's dependent round trips, beside which
			// the dictionary's one is not seen — same-box, every fused instantiation equal to the overlapped form it replaces, <4,4,1> at 2^30 with 0.1 % of patches
			// 81.0 / 81.2 / 78.7 against 81.2 / 78.9 / 81.0 % of the HBM peak, where the hook inside the chunk loop (one place, first chunk taken apart) gave 78.8 / 72.0 / 78.2.
			stage();
			translate<UncompressedSize, DictionaryIndexSize, true, AnyAlignment>(
				decompressed, compressed_input, length, [staged](key_type key) { return staged[key]; }, patches, chunk_staging, stream_tiles::nothing_to_do(), grid);
			return;
		}
		translate<UncompressedSize, DictionaryIndexSize, true, AnyAlignment>(
			decompressed, compressed_input, length, [staged](key_type key) { return staged[key]; }, patches, chunk_staging, stage, grid);
	}
	else if (cache_dictionary_in_shared_memory) {
		value_type* staged = reinterpret_cast<value_type*>(dynamic_lds);
		for (uint64_t i = threadIdx.x; i < num_dictionary_entries; i += blockDim.x) { staged[i] = dictionary_entries[i]; }
		__syncthreads();
		translate<UncompressedSize, DictionaryIndexSize, false, AnyAlignment>(
			decompressed, compressed_input, length, [staged](key_type key) { return staged[key]; }, patches, chunk_staging, stream_tiles::nothing_to_do(), grid);
	}
	else {
		// (through L2 / HBM: a tile's 16 look-ups per lane in flight together — round 5: they went out four at a time, each four waited for before their store)
		translate<UncompressedSize, DictionaryIndexSize, false, AnyAlignment, true>(
			decompressed, compressed_input, length, [dictionary_entries](key_type key) { return dictionary_entries[key]; }, patches, chunk_staging, stream_tiles::nothing_to_do(), grid);
	}
}

} // namespace detail

template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize>
__global__ void decompress(
	uint_t<UncompressedSize>*           __restrict__  decompressed,
	const uint_t<DictionaryIndexSize>*  __restrict__  compressed_input,
	const uint_t<UncompressedSize>*     __restrict__  dictionary_entries,
	size_type_by_index_size<IndexSize>                length,
	size_type_by_index_size<DictionaryIndexSize>      num_dictionary_entries,
	bool                                              cache_dictionary_in_shared_memory,
	serialization_factor_t                            serialization_factor) // accepted and ignored, as in gather.cuh:24
{
	(void) serialization_factor;
	detail::decompress<UncompressedSize, DictionaryIndexSize>(decompressed, compressed_input, dictionary_entries,
		length, num_dictionary_entries, cache_dictionary_in_shared_memory, sorted_patches::none());
}

/// The same, for a column handed over at an address that is aligned for its elements only (a slice [a, b) decoded into `decompressed + a`): 16-byte stores from the
/// output's first 16-byte boundary on, the keys' loads wide at whatever address that leaves them (stream_tiles.hpp, `AnyAlignment`).  The wrapper launches it when
/// either address is not as `decompress` wants it for its wide accesses — there that kernel goes one element at a time (26 % of the HBM peak against 88).
template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize>
__global__ void decompress_at_any_alignment(
	uint_t<UncompressedSize>*           __restrict__  decompressed,
	const uint_t<DictionaryIndexSize>*  __restrict__  compressed_input,
	const uint_t<UncompressedSize>*     __restrict__  dictionary_entries,
	size_type_by_index_size<IndexSize>                length,
	size_type_by_index_size<DictionaryIndexSize>      num_dictionary_entries,
	bool                                              cache_dictionary_in_shared_memory,
	serialization_factor_t                            serialization_factor)
{
	(void) serialization_factor;
	detail::decompress<UncompressedSize, DictionaryIndexSize, sorted_patches::none, true>(decompressed, compressed_input, dictionary_entries,
		length, num_dictionary_entries, cache_dictionary_in_shared_memory, sorted_patches::none());
}

/// A piece of a batch launch (giddy/batch.hpp; launched as batch::decode_pieces<piece_body<...>>): `decompress`'s parameters out of the piece's record,
/// its body on the piece's part of the grid.  `AnyAlignment`: the build for pieces at any element offset (the wrapper takes it for the whole batch when one piece asks).
template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize, bool AnyAlignment = false>
struct piece_body {
	enum : unsigned { max_threads_per_block = 1024 };
	static __device__ __forceinline__ void run(const batch::piece_view piece, const launch_grid::part grid)
	{
		detail::decompress<UncompressedSize, DictionaryIndexSize, sorted_patches::none, AnyAlignment>(
			piece.get<uint_t<UncompressedSize>*>(0), piece.get<const uint_t<DictionaryIndexSize>*>(1), piece.get<const uint_t<UncompressedSize>*>(2),
			piece.get<size_type_by_index_size<IndexSize>>(3), piece.get<size_type_by_index_size<DictionaryIndexSize>>(4), piece.get<bool>(5),
			sorted_patches::none(), false, grid);
	}
};

namespace patched {

/// Dictionary decompression and patching in one pass: what `dictionary::decompress` followed by
/// `data_layout::scatter::scatter(decompressed, patch_values, patch_positions, num_patches)` leaves in `decompressed`,
/// for ASCENDING patch positions (equal positions: the last one wins).
template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize, bool ManyPatches = false>
__global__ void decompress(
	uint_t<UncompressedSize>*           __restrict__  decompressed,
	const uint_t<DictionaryIndexSize>*  __restrict__  compressed_input,
	const uint_t<UncompressedSize>*     __restrict__  dictionary_entries,
	const uint_t<UncompressedSize>*     __restrict__  patch_values,
	const uint_t<IndexSize>*            __restrict__  patch_positions,
	size_type_by_index_size<IndexSize>                length,
	size_type_by_index_size<DictionaryIndexSize>      num_dictionary_entries,
	size_type_by_index_size<IndexSize>                num_patches,
	bool                                              cache_dictionary_in_shared_memory,
	bool                                              chunk_key_staging_in_dynamic_lds)   // room for detail::chunk_key_storage behind the dictionary
{
	__shared__ sorted_patches::window_storage<IndexSize, uint_t<UncompressedSize>> patch_windows;
	__shared__ sorted_patches::row_storage<uint_t<UncompressedSize>, ManyPatches ? 4 : 0, 16 / UncompressedSize, IndexSize> patch_rows;
	detail::decompress<UncompressedSize, DictionaryIndexSize>(decompressed, compressed_input, dictionary_entries,
		length, num_dictionary_entries, cache_dictionary_in_shared_memory,
		sorted_patches::cursor_t<IndexSize, uint_t<UncompressedSize>, ManyPatches>(patch_values, patch_positions, num_patches, length, patch_windows, patch_rows),
		chunk_key_staging_in_dynamic_lds);
}

} // namespace patched

template <unsigned IndexSize, unsigned UncompressedSize, unsigned DictionaryIndexSize>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;

	bool caches_dictionary = false;
	size_t staged_dictionary_bytes = 0;

	/// Room behind the dictionary for the wave-chunk shape's staging of narrow keys (stream_tiles::chunk_input_storage) and the
	/// residency that goes with it — 2 workgroups per CU: 80-83 % of HBM peak, 3 to 5: 70-73 (tools/occupancy_sweep.py).  Not
	/// for dictionaries staged once on a persistent grid (they keep the LDS) nor for look-ups through L2 (they want the occupancy).  (Asked for by the fused wrapper; the plain kernels, with the
	/// reference's parameter list, have no way to be told and stage nothing.)
	void reserve_chunk_key_staging()
	{
		if (!detail::stages_chunk_keys<UncompressedSize, DictionaryIndexSize>()) { return; }
		if (!caches_dictionary || staged_dictionary_bytes > GIDDY_DICT_RESTAGE_UP_TO_BYTES) { return; }
		if (DictionaryIndexSize > 1) {   // (1-byte keys: the kernel has static storage for it)
			size_t behind = (staged_dictionary_bytes + 15) / 16 * 16;
			size_t wanted = behind + sizeof(detail::chunk_key_storage<UncompressedSize, DictionaryIndexSize>);
			if (wanted > available_dynamic_shared_memory_per_block) { return; }
			dynamic_shared_memory_requirement.per_block = (shared_memory_size_t) wanted;
		}
		resident_blocks_per_multiprocessor_cap = 2u;
	}

	/// whether a dictionary of this many entries is staged in LDS (unless overridden)
	bool should_cache_dictionary_in_shared_memory(size_t num_dictionary_entries) const
	{
		return num_dictionary_entries * UncompressedSize <= available_dynamic_shared_memory_per_block;
	}

	launch_config_resolution_params_t(
		device::properties_t            device_properties_,
		device_function::attributes_t   kernel_function_attributes_,
		size_t                          num_dictionary_entries,
		size_t                          data_length,
		optional<bool>                  cache_dictionary_in_shared_memory_override = nullopt,
		optional<shared_memory_size_t>  dynamic_shared_mem_limit = nullopt)
	: parent(device_properties_, kernel_function_attributes_, dynamic_shared_mem_limit)
	{
		bool cache = cache_dictionary_in_shared_memory_override.value_or(
			should_cache_dictionary_in_shared_memory(num_dictionary_entries));
		// (the kernel takes wave chunks where it re-stages the dictionary per workgroup with a few entries per thread: its first branch)
		constexpr unsigned restaged_entries_per_thread = DictionaryIndexSize == 1 ? 1 : GIDDY_DICT_RESTAGE_ENTRIES_PER_THREAD;
		const bool restaged = cache && num_dictionary_entries <= (size_t) GIDDY_STREAM_THREADS_PER_BLOCK * restaged_entries_per_thread;
		const bool by_chunks = detail::by_wave_chunks<false, UncompressedSize, DictionaryIndexSize, false>() ||
			(restaged && detail::by_wave_chunks<false, UncompressedSize, DictionaryIndexSize, true>());
		const unsigned elements_per_thread_per_tile = by_chunks ? stream_tiles::elements_per_thread_per_chunk<uint_t<UncompressedSize>>()
			: restaged ? stream_tiles::elements_per_thread_per_tile<uint_t<UncompressedSize>, detail::steps_per_tile_for<UncompressedSize, DictionaryIndexSize, true>()>()
			: stream_tiles::elements_per_thread_per_tile<uint_t<UncompressedSize>>();
		grid_construction_resolution  = thread;
		length                        = util::div_rounding_up(data_length, elements_per_thread_per_tile);
		serialization_option          = auto_maximized;
		quanta.threads_in_block       = warp_size;
		upper_limits.threads_in_block = GIDDY_STREAM_THREADS_PER_BLOCK;
		// Dictionaries of up to 32 KiB: one tile per workgroup (tiles handed out in address order, which HBM likes
		// best — DESIGN.md §4), the dictionary re-staged for every tile from L2 with its request overlapped by the
		// tile's key loads (measured better than staging once on a persistent grid: 4096 x 4 bytes 66 -> 73 %).
		// A larger staged dictionary is amortised instead: a persistent grid of exactly as many workgroups as fit
		// on the chip with that much LDS each (at most 8 per CU), every one staging once and walking tiles at grid stride.
		keep_gpu_busy_factor          = 1024;
		size_t dictionary_bytes = num_dictionary_entries * UncompressedSize;
		if (cache && dictionary_bytes > GIDDY_DICT_RESTAGE_UP_TO_BYTES) {
			size_t lds_per_cu = device_properties.maxSharedMemoryPerMultiProcessor ? device_properties.maxSharedMemoryPerMultiProcessor : 65536;
			// (measured: five 32 KiB workgroups do NOT fit a 160 KiB CU — leave a 1 KiB margin)
			size_t workgroups_per_cu = std::min<size_t>(8, std::max<size_t>(1, (lds_per_cu - 1024) / dictionary_bytes));
			// fewer workgroups per CU -> larger workgroups, to keep ~32 wavefronts per CU in flight
			grid_block_dimension_t threads = workgroups_per_cu >= 8 ? 256 : (workgroups_per_cu >= 4 ? 512 : 1024);
			upper_limits.threads_in_block = threads;
			// the cap the resolver applies is CUs x 4 x factor / wavefronts-per-workgroup
			keep_gpu_busy_factor = (unsigned) std::max<size_t>(1, workgroups_per_cu * (threads / warp_size) / 4);
		}
		if (cache) { dynamic_shared_memory_requirement.per_block = (shared_memory_size_t) dictionary_bytes; }
		caches_dictionary = cache;
		staged_dictionary_bytes = dictionary_bytes;
		// 1-byte values: 3 resident workgroups per CU are 3-5 points faster than 8 (measured per instantiation, tools/occupancy_sweep.py)
		if (UncompressedSize == 1 && dictionary_bytes <= 2048) { resident_blocks_per_multiprocessor_cap = 3u; }
		if (UncompressedSize == 2 && dictionary_bytes <= 2048) { resident_blocks_per_multiprocessor_cap = 6u; }   // (+2..4 points)
		if (cache && UncompressedSize >= 4 && dictionary_bytes <= 2048 && GIDDY_DICT_WIDE_RESIDENT_BLOCKS_CAP) { resident_blocks_per_multiprocessor_cap = (unsigned) GIDDY_DICT_WIDE_RESIDENT_BLOCKS_CAP; }
		default_serialization_factor  = get_default_serialization_factor(cache);
	}

	launch_config_resolution_params_t(
		device::properties_t device_properties_, size_t num_dictionary_entries, size_t data_length,
		optional<bool> cache_dictionary_in_shared_memory_override = nullopt,
		optional<shared_memory_size_t> dynamic_shared_mem_limit = nullopt)
	: launch_config_resolution_params_t(device_properties_,
		device_function_t(decompress<IndexSize, UncompressedSize, DictionaryIndexSize>).attributes(),
		num_dictionary_entries, data_length, cache_dictionary_in_shared_memory_override, dynamic_shared_mem_limit) { }
};

} // namespace dictionary
} // namespace decompression
} // namespace kernels
} // namespace cuda
