// The tile loop every fixed-rate ("streaming") decoder shares: FOR, Dictionary, DZB-fixed, Model.
//
// A lane owns G = 16/sizeof(Out) consecutive outputs per step — one 16-byte store, contiguous across
// the wavefront — produced from G inputs read as one 1..16-byte load.  A tile is K steps of a whole
// workgroup (loads of all K steps are issued before the first store); tiles are taken at grid stride,
// and the launch parameters size the grid so that each workgroup normally gets exactly one tile
// (DESIGN.md §4).  Whatever the vector path cannot cover — a ragged tail, or everything when a base
// pointer is not aligned for vector access — goes through the same functor one element at a time.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "giddy/kernels/tuning.hpp"
#include "giddy/on_device/launch_grid.hpp"
#include "giddy/on_device/sorted_patches.hpp"
#include "giddy/on_device/wave64.hpp"

namespace cuda {
namespace stream_tiles {

enum : unsigned { steps_per_tile = GIDDY_STREAM_STEPS_PER_TILE };

template <typename Out> struct group_size { enum : unsigned { value = 16 / sizeof(Out) }; };

/// elements one thread produces per tile — what the launch parameters divide the column length by
template <typename Out, unsigned Steps = steps_per_tile> constexpr unsigned elements_per_thread_per_tile() { return group_size<Out>::value * Steps; }

struct nothing_to_do { __device__ __forceinline__ void operator()() const { } };

/// out[p] = produce(p, in[p]) for p < n.  `produce_group(first_position, inputs, outputs)` handles G
/// consecutive elements at once (it may hoist per-group work such as a segment lookup).
/// `after_loads()` is called by every thread of the workgroup EXACTLY ONCE and never inside a loop — behind the loads of the workgroup's
/// first tile (if it has one), before the first produce: the place for a side table's trip through LDS with its barrier, so that the
/// table's latency overlaps the tile's own loads.  In transform() there are two such places — the first tile whole, or not — chosen by a
/// condition that is the same for every thread of the workgroup (which is what HIP asks of a barrier in conditional code); in
/// transform_and_patch(), where wavefronts differ in their chunks, there is ONE.  (Until round 6 the hook ran per tile and per chunk,
/// guarded by a flag: a workgroup's wavefronts could arrive at different s_barrier instructions.)
struct no_context { };
struct nothing_to_prepare { __device__ __forceinline__ no_context operator()(uint64_t) const { return no_context(); } };

/// calls produce_group with or without the group's context, whichever it takes
template <typename ProduceGroup, typename Inputs, typename Outputs, typename Context>
__device__ __forceinline__ auto produce_with(ProduceGroup& produce_group, uint64_t position, const Inputs& inputs, Outputs& outputs, const Context& context)
	-> decltype(produce_group(position, inputs, outputs, context), void()) { produce_group(position, inputs, outputs, context); }
template <typename ProduceGroup, typename Inputs, typename Outputs>
__device__ __forceinline__ auto produce_with(ProduceGroup& produce_group, uint64_t position, const Inputs& inputs, Outputs& outputs, const no_context&)
	-> decltype(produce_group(position, inputs, outputs), void()) { produce_group(position, inputs, outputs); }

/// elements from `out` to the first 128-byte boundary at or after it (at most n): where the groups of an `AnyAlignment` build of transform() start
template <typename Out>
__host__ __device__ __forceinline__ uint64_t elements_before_first_line(const Out* out, uint64_t n)
{
	const uint64_t to_boundary = ((128 - (reinterpret_cast<uintptr_t>(out) & 127)) & 127) / sizeof(Out);
	return to_boundary < n ? to_boundary : n;
}

/// whether a launch should take the `AnyAlignment` build of a transform() decoder: an output that does not start a 128-byte line, or an input not aligned for a group's load
template <typename In, typename Out>
__host__ __forceinline__ bool wants_any_alignment_build(const Out* out, const In* in)
{
	constexpr unsigned G = 16 / sizeof(Out);
	constexpr unsigned in_bytes = G * sizeof(In) >= 16 ? 16 : G * sizeof(In);
	return (reinterpret_cast<uintptr_t>(out) & 127) != 0 || (reinterpret_cast<uintptr_t>(in) & (in_bytes - 1)) != 0;
}

/// transform()'s tile loop for a decoder with an `after_loads` hook, arranged so that the hook is called from one place: the first tile's loads, the hook, the first
/// tile's produces and stores — then the workgroup's other tiles as transform() itself takes them.  Every quantity the loop branches on (the tile index, whether a
/// tile is whole) is the same for all the threads of a workgroup.  `lead`: elements in front of the first group (the AnyAlignment build), else 0.
template <typename In, typename Out, unsigned K, int Stores, bool AnyAlignment, typename ProduceGroup, typename AfterLoads, typename Grid>
__device__ __forceinline__ void pipelined_tiles(
	Out* __restrict__ out, const In* __restrict__ in, uint64_t lead, uint64_t num_groups, ProduceGroup& produce_group, AfterLoads& after_loads, const Grid& grid)
{
	using namespace memory_access;
	constexpr unsigned G = group_size<Out>::value;
	const uint64_t groups_per_tile = (uint64_t) blockDim.x * K;
	group<In, G> inputs[K];
	auto load = [&](uint64_t g) { if constexpr (AnyAlignment) { return load_group_at_any_alignment<In, G>(in + g * G); } else { return load_group<In, G>(in + g * G); } };
	auto load_tile = [&](uint64_t tile) {
		const uint64_t first_group = tile * groups_per_tile + threadIdx.x;
		if ((tile + 1) * groups_per_tile <= num_groups) {
			#pragma unroll
			for (unsigned k = 0; k < K; k++) { inputs[k] = load(first_group + (uint64_t) k * blockDim.x); }
		}
		else {
			#pragma unroll
			for (unsigned k = 0; k < K; k++) {
				const uint64_t g = first_group + (uint64_t) k * blockDim.x;
				if (g < num_groups) { inputs[k] = load(g); }
			}
		}
	};
	// ONLY THE FIRST TILE is taken apart around the hook.  The tiles behind it — a workgroup has them when the column has more tiles than the grid may have workgroups:
	// 8-byte values from 2^30 elements on — keep loads, produces and stores in ONE block each: with every tile's loads issued at the end of the trip before (the first
	// version of this loop) they sat behind a branch, the wait for them became vmcnt(0), and that waits for the previous tile's write-through stores as well
	// (Dictionary <*,8,1> at 2^30: 76 -> 62 % of the HBM peak; at 2^28, one tile per workgroup, nothing showed).
	auto produce_and_store_tile = [&](uint64_t tile) {
		const uint64_t first_group = tile * groups_per_tile + threadIdx.x;
		if ((tile + 1) * groups_per_tile <= num_groups) {
			// a whole tile (every tile but possibly the last): no per-step bounds checks
			#pragma unroll
			for (unsigned k = 0; k < K; k++) {
				const uint64_t g = first_group + (uint64_t) k * blockDim.x;
				group<Out, G> outputs;
				produce_group(lead + g * G, inputs[k], outputs);
				store_group<Stores>(out + g * G, outputs);
				// narrow elements: G look-ups per step — keep the steps' work from being hoisted above one another
				// (all K x 16 look-ups of a 1-byte column in flight at once cost half the occupancy)
				if (G >= 8) { __builtin_amdgcn_sched_barrier(0); }
			}
		}
		else {
			#pragma unroll
			for (unsigned k = 0; k < K; k++) {
				const uint64_t g = first_group + (uint64_t) k * blockDim.x;
				if (g >= num_groups) { continue; }
				group<Out, G> outputs;
				produce_group(lead + g * G, inputs[k], outputs);
				store_group<Stores>(out + g * G, outputs);
			}
		}
	};
	uint64_t tile = grid.block();
	const bool any = tile * groups_per_tile < num_groups;   // (the same for every thread of the workgroup, as is `whole` below: they come from blockIdx and the column's length)
	if (any && (tile + 1) * groups_per_tile <= num_groups) {
		// the usual case — the workgroup's first tile is a whole one — in ONE block: loads, hook, produces and stores.  (With the loads inside a branch of their own and the
		// hook behind its join, narrow inputs — 1-byte keys of 8-byte values, a 2-byte load per step that is unpacked into its group — were waited for and unpacked BEFORE
		// the hook's barrier: the overlap of the two round trips this hook exists for was gone, Dictionary <*,8,1> 76 -> 62-68 % of the HBM peak at 2^30.)
		const uint64_t first_group = tile * groups_per_tile + threadIdx.x;
		#pragma unroll
		for (unsigned k = 0; k < K; k++) { inputs[k] = load(first_group + (uint64_t) k * blockDim.x); }
		after_loads();
		#pragma unroll
		for (unsigned k = 0; k < K; k++) {
			const uint64_t g = first_group + (uint64_t) k * blockDim.x;
			group<Out, G> outputs;
			produce_group(lead + g * G, inputs[k], outputs);
			store_group<Stores>(out + g * G, outputs);
			if (G >= 8) { __builtin_amdgcn_sched_barrier(0); }
		}
	}
	else {
		// (no tile for this workgroup — the hook still runs: the column's scalar tail looks things up too —, or a ragged one)
		if (any) { load_tile(tile); }
		after_loads();
		if (any) { produce_and_store_tile(tile); }
	}
	if (!any) { return; }
	for (tile += grid.blocks(); tile * groups_per_tile < num_groups; tile += grid.blocks()) {
		const uint64_t first_group = tile * groups_per_tile + threadIdx.x;
		if ((tile + 1) * groups_per_tile <= num_groups) {
			#pragma unroll
			for (unsigned k = 0; k < K; k++) { inputs[k] = load(first_group + (uint64_t) k * blockDim.x); }
			#pragma unroll
			for (unsigned k = 0; k < K; k++) {
				const uint64_t g = first_group + (uint64_t) k * blockDim.x;
				group<Out, G> outputs;
				produce_group(lead + g * G, inputs[k], outputs);
				store_group<Stores>(out + g * G, outputs);
				if (G >= 8) { __builtin_amdgcn_sched_barrier(0); }
			}
		}
		else {
			load_tile(tile);
			produce_and_store_tile(tile);
		}
	}
}

/// `prepare(first_position)` -> a group's context (say, its segment's record): for a whole tile ALL the contexts are
/// requested right behind the tile's input loads and before its first store — a load issued between two stores would
/// have to wait for the earlier store (CDNA4 counts vector memory operations in order).
/// `Stores`: memory_access::store_policy of the tile's 16-byte stores
/// `AnyAlignment`: a build for arrays aligned for their elements only (a slice [a, b) of a column decoded into `out + a`): the groups start at the first 128-byte
/// boundary of the OUTPUT — the elements before it go one at a time with the column's last ones — and the input's loads are wide at whatever address that leaves
/// them.  A build of its own, chosen by the wrapper from the addresses: inside the one kernel the extra address arithmetic cost the aligned column 2 points.
/// (only for a produce_group that does not care where a group starts: positions are passed as they are in the column, but a group may lie across any boundary)
/// `ProduceAllFirst`: a whole tile's K groups are produced before the first of them is stored — for a produce_group that LOADS (a dictionary looked up through L2 / HBM): its
/// K x G look-ups are then in flight together, where produce-store-produce-store waits for each group's (the stores are inline assembly the compiler moves nothing across).
template <typename In, typename Out, unsigned Steps = steps_per_tile, int Stores = memory_access::non_temporal, bool AnyAlignment = false, bool ProduceAllFirst = false,
	typename ProduceGroup, typename ProduceOne, typename AfterLoads = nothing_to_do, typename Prepare = nothing_to_prepare, typename Grid = launch_grid::whole>
__device__ __forceinline__ void transform(
	Out* __restrict__ out, const In* __restrict__ in, uint64_t n, ProduceGroup produce_group, ProduceOne produce_one,
	AfterLoads after_loads = AfterLoads(), Prepare prepare = Prepare(), Grid grid = Grid())   // (`grid`: launch_grid.hpp — the launch's own, or a batch piece's part of it)
{
	using namespace memory_access;
	constexpr unsigned G = group_size<Out>::value;
	constexpr unsigned K = Steps;
	constexpr unsigned in_bytes = G * sizeof(In) >= 16 ? 16 : G * sizeof(In);
	constexpr bool has_hook = !std::is_same<AfterLoads, nothing_to_do>::value;
	static_assert(!has_hook || (std::is_same<Prepare, nothing_to_prepare>::value && !ProduceAllFirst), "a decoder with an after_loads hook takes the pipelined loop, which has neither contexts nor ProduceAllFirst");

	if constexpr (AnyAlignment) {
		// (the first 128-BYTE boundary: a wavefront's store instruction then writes whole lines — from a 16-byte boundary that is not one of those, the same tiles run at 54 % of the HBM peak against 91)
		const uint64_t lead = elements_before_first_line(out, n);
		Out* __restrict__ out_at_boundary = out + lead;
		const In* __restrict__ in_at_boundary = in + lead;
		const uint64_t num_groups = (n - lead) / G;
		const uint64_t groups_per_tile = (uint64_t) blockDim.x * K;
		if constexpr (has_hook) {
			pipelined_tiles<In, Out, K, Stores, true>(out_at_boundary, in_at_boundary, lead, num_groups, produce_group, after_loads, grid);
		}
		else
		for (uint64_t tile = grid.block(); tile * groups_per_tile < num_groups; tile += grid.blocks()) {
			const uint64_t first_group = tile * groups_per_tile + threadIdx.x;
			group<In, G> inputs[K];
			#pragma unroll
			for (unsigned k = 0; k < K; k++) {
				const uint64_t g = first_group + (uint64_t) k * blockDim.x;
				if (g < num_groups) { inputs[k] = load_group_at_any_alignment<In, G>(in_at_boundary + g * G); }
			}
			#pragma unroll
			for (unsigned k = 0; k < K; k++) {
				const uint64_t g = first_group + (uint64_t) k * blockDim.x;
				if (g >= num_groups) { continue; }
				group<Out, G> outputs;
				produce_with(produce_group, lead + g * G, inputs[k], outputs, prepare(lead + g * G));
				store_group<Stores>(out_at_boundary + g * G, outputs);
			}
		}
		const uint64_t global_thread = (uint64_t) grid.block() * blockDim.x + threadIdx.x;
		const uint64_t num_threads = (uint64_t) grid.blocks() * blockDim.x;
		for (uint64_t p = global_thread; p < lead; p += num_threads) { out[p] = produce_one(p, in[p]); }
		for (uint64_t p = lead + num_groups * G + global_thread; p < n; p += num_threads) { out[p] = produce_one(p, in[p]); }
		return;
	}

	const bool vectorizable = is_aligned(out, 16) && is_aligned(in, in_bytes);
	const uint64_t num_groups = vectorizable ? n / G : 0;
	const uint64_t groups_per_tile = (uint64_t) blockDim.x * K;

	if constexpr (has_hook) {
		pipelined_tiles<In, Out, K, Stores, false>(out, in, 0, num_groups, produce_group, after_loads, grid);
	}
	else
	for (uint64_t tile = grid.block(); tile * groups_per_tile < num_groups; tile += grid.blocks()) {
		const uint64_t first_group = tile * groups_per_tile + threadIdx.x;
		group<In, G> inputs[K];
		if ((tile + 1) * groups_per_tile <= num_groups) {
			// a whole tile (every tile but possibly the last): no per-step bounds checks, so the K loads and the
			// K stores form straight-line code the compiler can schedule with counted waits
			#pragma unroll
			for (unsigned k = 0; k < K; k++) { inputs[k] = load_group<In, G>(in + (first_group + (uint64_t) k * blockDim.x) * G); }
			decltype(prepare((uint64_t) 0)) contexts[K];
			#pragma unroll
			for (unsigned k = 0; k < K; k++) { contexts[k] = prepare((first_group + (uint64_t) k * blockDim.x) * G); }
			if constexpr (ProduceAllFirst) {
				group<Out, G> outputs[K];
				#pragma unroll
				for (unsigned k = 0; k < K; k++) { produce_with(produce_group, (first_group + (uint64_t) k * blockDim.x) * G, inputs[k], outputs[k], contexts[k]); }
				#pragma unroll
				for (unsigned k = 0; k < K; k++) { store_group<Stores>(out + (first_group + (uint64_t) k * blockDim.x) * G, outputs[k]); }
			}
			else {
			#pragma unroll
			for (unsigned k = 0; k < K; k++) {
				const uint64_t g = first_group + (uint64_t) k * blockDim.x;
				group<Out, G> outputs;
				produce_with(produce_group, g * G, inputs[k], outputs, contexts[k]);
				store_group<Stores>(out + g * G, outputs);
				// narrow elements: G look-ups per step — keep the steps' work from being hoisted above one another
				// (all K x 16 look-ups of a 1-byte column in flight at once cost half the occupancy)
				if (G >= 8) { __builtin_amdgcn_sched_barrier(0); }
			}
			}
#if defined(GIDDY_STREAM_STAMP_PROBE)
			// (experiment: when was this tile finished, and on which XCD — in place of the tile's first 8 bytes of output; tools/stream_stamp_probe.py)
			if (threadIdx.x == 0) {
				*reinterpret_cast<uint64_t*>(out + tile * groups_per_tile * G) = (__builtin_amdgcn_s_memrealtime() << 4) | (__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15);
			}
#endif
			continue;
		}
		#pragma unroll
		for (unsigned k = 0; k < K; k++) {
			uint64_t g = first_group + (uint64_t) k * blockDim.x;
			if (g < num_groups) { inputs[k] = load_group<In, G>(in + g * G); }
		}
		#pragma unroll
		for (unsigned k = 0; k < K; k++) {
			uint64_t g = first_group + (uint64_t) k * blockDim.x;
			if (g >= num_groups) { continue; }
			group<Out, G> outputs;
			produce_with(produce_group, g * G, inputs[k], outputs, prepare(g * G));
			store_group<Stores>(out + g * G, outputs);
		}
	}

	const uint64_t global_thread = (uint64_t) grid.block() * blockDim.x + threadIdx.x;
	const uint64_t num_threads = (uint64_t) grid.blocks() * blockDim.x;
	for (uint64_t p = num_groups * G + global_thread; p < n; p += num_threads) { out[p] = produce_one(p, in[p]); }
}

// ---- the same transformation with patches merged in -----------------------------------------
//
// Finding a tile's first patch costs dependent round trips to memory; a workgroup that lives for one small tile
// cannot hide them (measured on MI355X, Dictionary <4,4,1>, 0.1 % patches: +14..31 % per tile-wise variant tried,
// against +4 % for a separate scatter launch).  So the patched decoders run the Delta way instead: a WAVEFRONT
// takes a chunk of `rows_per_chunk` consecutive rows of 64*G outputs, seeks once while its first rows are in
// flight, and walks its rows with the next rows' loads issued before the current rows' stores.  Measured for
// chunks of 4..64 rows with 2..8 rows in flight: 8 rows, 4 in flight decodes Dictionary <4,4,1> WITH 0.1 % patches
// as fast as the tile loop above decodes it without (0.87-0.90 ms for 2^30 elements); 32 rows / 2 in flight
// — the Delta shape — is 8 % slower, 4 rows 17-40 % slower.

enum : unsigned { rows_in_flight = 4 };

/// rows a wavefront takes per chunk: 8 — except for 8-byte outputs, whose rows hold only 128 elements (measured at 8, 16 and 32
/// rows: Dictionary <4,8,1> 40 / 47 / 51 % of peak, <4,8,2> 58 / 66 / 69 %; 4-byte outputs lose 5-9 points going from 8 to 16+)
template <typename Out> constexpr unsigned rows_per_chunk_for()
{
	return sizeof(Out) >= 8 ? 32 : 8;
}

/// elements one thread produces per chunk — what the patched launch parameters divide the column length by
template <typename Out> constexpr unsigned elements_per_thread_per_chunk() { return group_size<Out>::value * rows_per_chunk_for<Out>(); }

/// Rows of at most 2 input bytes per lane (1-byte inputs of 8-byte outputs): a wavefront-wide load instruction would fetch
/// 128 bytes.  Given a chunk_input_storage (wave-private LDS, declared __shared__ by the kernel), a chunk's input comes in
/// with 16-byte loads instead and is dealt out to the rows through LDS (as in Delta's staged-input path).
template <typename In, typename Out, unsigned UpToBytesPerLane = 2> constexpr bool stages_chunk_input() { return group_size<Out>::value * sizeof(In) <= UpToBytesPerLane; }
enum : unsigned { chunk_staging_waves = 4 };   // wavefronts per workgroup the storage is provisioned for; others load row by row
template <typename In, typename Out, bool Enabled = stages_chunk_input<In, Out>()>
struct alignas(16) chunk_input_storage {
	enum : unsigned { elements = rows_per_chunk_for<Out>() * wave64::size * group_size<Out>::value };
	In inputs[Enabled ? chunk_staging_waves : 1][Enabled ? elements : 1];
};

/// The rows of a WHOLE chunk in the row form (rows_per_chunk rows, the first R of them already requested into `current`), as straight-line code: no bound in
/// sight, every load and store of the chunk unconditional, so that the compiler can wait for a row's input by COUNT (CDNA4 counts vector memory operations in
/// order: `s_waitcnt vmcnt(n)` with the newer requests still in flight).  The bounded loop the ragged last chunk takes wraps every load and store in a
/// wave-uniform branch; behind a branch the count is unknown and every wait becomes vmcnt(0) — which waits for the next rows' loads just issued AND for the
/// stores before (they are counted too): a chunk was eight serialized round trips (round 6, read off the ISA: plain Dictionary <4,8,2> 61 -> 76 % of the HBM peak).
/// (All 32 rows of an 8-byte chunk spelled out cost 15-18 registers — addresses kept for every row — and a spill in fused Dictionary <4,8,8>: hence pairs around a loop.)
template <typename In, typename Out, unsigned R, typename ProduceGroup, typename Patches>
__device__ __forceinline__ void rows_of_a_whole_chunk(
	Out* __restrict__ out, const In* __restrict__ in, uint64_t first_row, unsigned lane, memory_access::group<In, group_size<Out>::value> (&current)[R],
	ProduceGroup& produce_group, Patches& patches)
{
	using namespace memory_access;
	constexpr unsigned G = group_size<Out>::value;
	constexpr unsigned rows_per_chunk = rows_per_chunk_for<Out>();
	constexpr unsigned row = wave64::size * G;
	static_assert(rows_per_chunk % R == 0, "a whole chunk is a whole number of groups of R rows");
	constexpr unsigned groups = rows_per_chunk / R;   // groups of R rows: group g is requested while group g - 1 is produced and stored
	auto request = [&](group<In, G> (&into)[R], unsigned g) {
		#pragma unroll
		for (unsigned k = 0; k < R; k++) { into[k] = load_group<In, G>(in + (first_row + g * R + k) * row + lane * G); }
	};
	auto produce_and_store = [&](const group<In, G> (&from)[R], unsigned g) {
		#pragma unroll
		for (unsigned k = 0; k < R; k++) {
			const uint64_t r = first_row + g * R + k;
			const uint64_t p = r * row + lane * G;
			group<Out, G> outputs;
			produce_group(p, from[k], outputs);
			patches.template apply_to_row<G>(r * row, lane, outputs);
			store_group(out + p, outputs);
			// narrow elements: G look-ups per row — keep the rows' look-ups from being hoisted above one another (all 4 x 16 of a 1-byte column in flight at once spill)
			if (G >= 8) { __builtin_amdgcn_sched_barrier(0); }
		}
	};
	group<In, G> other[R];
	if constexpr (groups == 1) { produce_and_store(current, 0); }
	else if constexpr (groups % 2 != 0 || groups == 2) {
		// (few groups: all of them spelled out, the two buffers taking turns)
		#pragma unroll
		for (unsigned g = 0; g < groups; g++) {
			if (g % 2 == 0) { if (g + 1 < groups) { request(other, g + 1); } produce_and_store(current, g); }
			else { if (g + 1 < groups) { request(current, g + 1); } produce_and_store(other, g); }
		}
	}
	else {
		// PAIRS of groups around a loop, the two buffers taking turns with no copy between them (a copy of a buffer would wait for its loads); the last pair, which
		// requests nothing behind it, spelled out after the loop — so nothing inside the loop is conditional either
		#pragma unroll 1
		for (unsigned g = 0; g + 2 < groups; g += 2) {
			request(other, g + 1);
			produce_and_store(current, g);
			request(current, g + 2);
			produce_and_store(other, g + 1);
		}
		request(other, groups - 1);
		produce_and_store(current, groups - 2);
		produce_and_store(other, groups - 1);
	}
}

template <typename In, typename Out, unsigned RowsInFlight = rows_in_flight, unsigned StageUpToBytesPerLane = 2,
	typename ProduceGroup, typename ProduceOne, typename Patches, typename AfterLoads = nothing_to_do, typename Grid = launch_grid::whole>
__device__ __forceinline__ void transform_and_patch(
	Out* __restrict__ out, const In* __restrict__ in, uint64_t n, ProduceGroup produce_group, ProduceOne produce_one, Patches patches,
	AfterLoads after_loads = AfterLoads(),   // (as in transform(): called once per thread, from one place outside the chunk loop)
	chunk_input_storage<In, Out, stages_chunk_input<In, Out, StageUpToBytesPerLane>()>* staging = nullptr, Grid grid = Grid())
{
	using namespace memory_access;
	constexpr unsigned G = group_size<Out>::value;
	constexpr unsigned R = RowsInFlight;
	constexpr unsigned rows_per_chunk = rows_per_chunk_for<Out>();
	constexpr unsigned row = wave64::size * G;
	constexpr unsigned in_bytes = G * sizeof(In) >= 16 ? 16 : G * sizeof(In);

	const bool vectorizable = is_aligned(out, 16) && is_aligned(in, in_bytes);
	const uint64_t num_rows = vectorizable ? n / row : 0;   // whole rows; the rest goes one element at a time
	const uint64_t num_chunks = (num_rows + rows_per_chunk - 1) / rows_per_chunk;
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) grid.block() * wave64::per_block() + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) grid.blocks() * wave64::per_block();

	constexpr bool stage_chunk_input = stages_chunk_input<In, Out, StageUpToBytesPerLane>();
	constexpr unsigned chunk_input_bytes = rows_per_chunk * row * sizeof(In);
	const bool may_stage = stage_chunk_input && staging != nullptr && wave64::index_in_block() < chunk_staging_waves && is_aligned(in, 16);

	// (rows_of_a_whole_chunk for decoders WITHOUT patches only — the plain Dictionary of 8-byte values: 61 -> 76 % of the HBM peak.  With a cursor of patches in the rows the
	// waits are conservative behind its branches either way, and the straight-line form measured slower where rows carry many patches: fused FOR <4,4,2,linear> -8 points and
	// fused DZB-fixed <4,2,1> -6 from 2 % of patches on, nothing gained below: profiles/r06_fused_regret.txt, first table against the second)
	constexpr bool whole_chunks_in_straight_line = !Patches::enabled;
	constexpr bool has_hook = !std::is_same<AfterLoads, nothing_to_do>::value;
	if constexpr (has_hook) {
		// The same chunk loop arranged so that the hook (a side table's trip through LDS and its BARRIER) is called from one place that every wavefront of the
		// workgroup passes exactly once, whatever its number of chunks and whichever of the two forms its chunks take: the first chunk's seek probes and loads,
		// the hook, the first chunk's rows — then the wavefront's other chunks.
		using piece_t = typename chunk<16>::type;
		constexpr unsigned pieces_per_lane = stage_chunk_input ? chunk_input_bytes / 16 / wave64::size : 1;
		static_assert(!stage_chunk_input || (pieces_per_lane >= 1 && chunk_input_bytes % (16 * wave64::size) == 0), "a chunk's input is a whole number of 16-byte pieces per lane");
		piece_t pieces[pieces_per_lane];
		group<In, G> current[R], next[R];
		auto end_of = [&](uint64_t first_row) { return first_row + rows_per_chunk < num_rows ? first_row + rows_per_chunk : num_rows; };
		auto comes_through_lds = [&](uint64_t first_row) { return stage_chunk_input && may_stage && end_of(first_row) - first_row == rows_per_chunk; };   // wave-uniform
		auto issue = [&](uint64_t c) {
			const uint64_t first_row = c * rows_per_chunk, end_row = end_of(first_row);
			patches.begin_seek(first_row * row, lane);   // (its probes go out ahead of the first rows' loads, the rest behind them)
			if (comes_through_lds(first_row)) {
				const piece_t* source = reinterpret_cast<const piece_t*>(in + first_row * row);
				#pragma unroll
				for (unsigned i = 0; i < pieces_per_lane; i++) { pieces[i] = source[i * wave64::size + lane]; }
			}
			else if (whole_chunks_in_straight_line && end_row - first_row == rows_per_chunk) {   // (a whole chunk: nothing conditional about its loads — rows_of_a_whole_chunk)
				#pragma unroll
				for (unsigned k = 0; k < R; k++) { current[k] = load_group<In, G>(in + (first_row + k) * row + lane * G); }
			}
			else {
				#pragma unroll
				for (unsigned k = 0; k < R; k++) {
					if (first_row + k < end_row) { current[k] = load_group<In, G>(in + (first_row + k) * row + lane * G); }
				}
			}
		};
		// ONLY THE FIRST CHUNK is taken apart around the hook; a wavefront's other chunks (a column of more chunks than the grid may have wavefronts) request their rows at
		// the top of their own trip, as the loop below this one does — requested at the end of the trip before, the loads sit behind a branch and the wait for them also
		// waits for the previous chunk's stores (pipelined_tiles, above).
		auto rows_of_the_chunk = [&](uint64_t c) {
				const uint64_t first_row = c * rows_per_chunk, end_row = end_of(first_row);
				patches.finish_seek(lane);
				if (comes_through_lds(first_row)) {
					if constexpr (stage_chunk_input) {
						In* mine = staging->inputs[wave64::index_in_block()];
						wave64::lds_fence();   // (the previous chunk's rows have been read)
						#pragma unroll
						for (unsigned i = 0; i < pieces_per_lane; i++) { reinterpret_cast<piece_t*>(mine)[i * wave64::size + lane] = pieces[i]; }
						wave64::lds_fence();
						for (uint64_t r = first_row; r < end_row; r++) {
							const uint64_t p = r * row + lane * G;
							const group<In, G> inputs = load_group<In, G>(mine + (r - first_row) * row + lane * G);
							group<Out, G> outputs;
							produce_group(p, inputs, outputs);
							patches.template apply_to_row<G>(r * row, lane, outputs);
							store_group(out + p, outputs);
						}
					}
				}
				else if (whole_chunks_in_straight_line && end_row - first_row == rows_per_chunk) {   // wave-uniform
					rows_of_a_whole_chunk<In, Out, R>(out, in, first_row, lane, current, produce_group, patches);
				}
				else {
					for (uint64_t r = first_row; r < end_row; r += R) {
						#pragma unroll
						for (unsigned k = 0; k < R; k++) {
							if (r + R + k < end_row) { next[k] = load_group<In, G>(in + (r + R + k) * row + lane * G); }
						}
						#pragma unroll
						for (unsigned k = 0; k < R; k++) {
							if (r + k >= end_row) { continue; }   // wave-uniform
							const uint64_t p = (r + k) * row + lane * G;
							group<Out, G> outputs;
							produce_group(p, current[k], outputs);
							patches.template apply_to_row<G>((r + k) * row, lane, outputs);
							store_group(out + p, outputs);
							if (G >= 8) { __builtin_amdgcn_sched_barrier(0); }   // (as below)
						}
						#pragma unroll
						for (unsigned k = 0; k < R; k++) { current[k] = next[k]; }
					}
				}
		};
		if (wave < num_chunks) { issue(wave); }
		after_loads();
		if (wave < num_chunks) {
			rows_of_the_chunk(wave);
			for (uint64_t c = wave + num_waves; c < num_chunks; c += num_waves) { issue(c); rows_of_the_chunk(c); }
		}
	}
	else
	for (uint64_t c = wave; c < num_chunks; c += num_waves) {
		const uint64_t first_row = c * rows_per_chunk;
		const uint64_t end_row = first_row + rows_per_chunk < num_rows ? first_row + rows_per_chunk : num_rows;
		if constexpr (stage_chunk_input) {
			if (may_stage && end_row - first_row == rows_per_chunk) {   // wave-uniform
				using piece_t = typename chunk<16>::type;
				constexpr unsigned pieces_per_lane = chunk_input_bytes / 16 / wave64::size;
				static_assert(pieces_per_lane >= 1 && chunk_input_bytes % (16 * wave64::size) == 0, "a chunk's input is a whole number of 16-byte pieces per lane");
				In* mine = staging->inputs[wave64::index_in_block()];
				const piece_t* source = reinterpret_cast<const piece_t*>(in + first_row * row);
				piece_t pieces[pieces_per_lane];
				patches.begin_seek(first_row * row, lane);
				#pragma unroll
				for (unsigned i = 0; i < pieces_per_lane; i++) { pieces[i] = source[i * wave64::size + lane]; }
				patches.finish_seek(lane);
				wave64::lds_fence();   // (the previous chunk's rows have been read)
				#pragma unroll
				for (unsigned i = 0; i < pieces_per_lane; i++) { reinterpret_cast<piece_t*>(mine)[i * wave64::size + lane] = pieces[i]; }
				wave64::lds_fence();
				for (uint64_t r = first_row; r < end_row; r++) {
					const uint64_t p = r * row + lane * G;
					const group<In, G> inputs = load_group<In, G>(mine + (r - first_row) * row + lane * G);
					group<Out, G> outputs;
					produce_group(p, inputs, outputs);
					patches.template apply_to_row<G>(r * row, lane, outputs);
					store_group(out + p, outputs);
				}
				continue;
			}
		}
		group<In, G> current[R], next[R];
		patches.begin_seek(first_row * row, lane);   // (its probes go out ahead of the first rows' loads, the rest behind them)
		if (whole_chunks_in_straight_line && end_row - first_row == rows_per_chunk) {   // wave-uniform: a whole chunk, nothing conditional about its loads and stores
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { current[k] = load_group<In, G>(in + (first_row + k) * row + lane * G); }
			patches.finish_seek(lane);
			rows_of_a_whole_chunk<In, Out, R>(out, in, first_row, lane, current, produce_group, patches);
			continue;
		}
		#pragma unroll
		for (unsigned k = 0; k < R; k++) {
			if (first_row + k < end_row) { current[k] = load_group<In, G>(in + (first_row + k) * row + lane * G); }
		}
		patches.finish_seek(lane);
		for (uint64_t r = first_row; r < end_row; r += R) {
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				if (r + R + k < end_row) { next[k] = load_group<In, G>(in + (r + R + k) * row + lane * G); }
			}
			#pragma unroll
			for (unsigned k = 0; k < R; k++) {
				if (r + k >= end_row) { continue; }   // wave-uniform
				const uint64_t p = (r + k) * row + lane * G;
				group<Out, G> outputs;
				produce_group(p, current[k], outputs);
				patches.template apply_to_row<G>((r + k) * row, lane, outputs);
				store_group(out + p, outputs);
				// narrow elements: G look-ups per row — keep the rows' look-ups from being hoisted above one another
				// (all 4 x 16 of a 1-byte column in flight at once spill)
				if (G >= 8) { __builtin_amdgcn_sched_barrier(0); }
			}
			#pragma unroll
			for (unsigned k = 0; k < R; k++) { current[k] = next[k]; }
		}
	}

	const uint64_t global_thread = (uint64_t) grid.block() * blockDim.x + threadIdx.x;
	const uint64_t num_threads = (uint64_t) grid.blocks() * blockDim.x;
	for (uint64_t p = num_rows * row + global_thread; p < n; p += num_threads) { out[p] = patches.patched(p, produce_one(p, in[p])); }
}

/// out[p] = produce(p) with patches merged in: the wave-chunk loop of transform_and_patch() for a decoder with no input
/// array (Model) — a wavefront seeks once per chunk of RowsPerChunk rows and then only produces, patches and stores.
template <typename Out, unsigned RowsPerChunk, typename ProduceGroup, typename ProduceOne, typename Patches>
__device__ __forceinline__ void generate_and_patch(Out* __restrict__ out, uint64_t n, ProduceGroup produce_group, ProduceOne produce_one, Patches patches)
{
	using namespace memory_access;
	constexpr unsigned G = group_size<Out>::value;
	constexpr unsigned rows_per_chunk = RowsPerChunk;
	constexpr unsigned row = wave64::size * G;
	const uint64_t num_rows = is_aligned(out, 16) ? n / row : 0;   // whole rows; the rest goes one element at a time
	const uint64_t num_chunks = (num_rows + rows_per_chunk - 1) / rows_per_chunk;
	const unsigned lane = wave64::lane_id();
	const uint64_t wave = (uint64_t) blockIdx.x * wave64::per_block() + wave64::index_in_block();
	const uint64_t num_waves = (uint64_t) gridDim.x * wave64::per_block();

	for (uint64_t c = wave; c < num_chunks; c += num_waves) {
		const uint64_t first_row = c * rows_per_chunk;
		const uint64_t end_row = first_row + rows_per_chunk < num_rows ? first_row + rows_per_chunk : num_rows;
		patches.seek(first_row * row, lane);
		for (uint64_t r = first_row; r < end_row; r++) {
			const uint64_t p = r * row + lane * G;
			group<Out, G> outputs;
			produce_group(p, outputs);
			patches.template apply_to_row<G>(r * row, lane, outputs);
			store_group(out + p, outputs);
		}
	}
	const uint64_t global_thread = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t num_threads = (uint64_t) gridDim.x * blockDim.x;
	for (uint64_t p = num_rows * row + global_thread; p < n; p += num_threads) { out[p] = patches.patched(p, produce_one(p)); }
}

/// out[p] = produce(p): the same loop for a decoder with no input array (Model)
/// ProduceAllFirst: a produce_group that loads from memory (a per-interval record) runs for all K steps of a whole tile
/// before the tile's first store — a load issued between two stores would wait for the earlier store.
template <typename Out, bool ProduceAllFirst = false, unsigned Steps = steps_per_tile, int Stores = memory_access::non_temporal, typename ProduceGroup, typename ProduceOne>
__device__ __forceinline__ void generate(Out* __restrict__ out, uint64_t n, ProduceGroup produce_group, ProduceOne produce_one)
{
	using namespace memory_access;
	constexpr unsigned G = group_size<Out>::value;
	constexpr unsigned K = Steps;
	const uint64_t num_groups = is_aligned(out, 16) ? n / G : 0;
	const uint64_t groups_per_tile = (uint64_t) blockDim.x * K;
	for (uint64_t tile = blockIdx.x; tile * groups_per_tile < num_groups; tile += gridDim.x) {
		const uint64_t first_group = tile * groups_per_tile + threadIdx.x;
		if (ProduceAllFirst && (tile + 1) * groups_per_tile <= num_groups) {
			group<Out, G> outputs[K];
			#pragma unroll
			for (unsigned k = 0; k < K; k++) { produce_group((first_group + (uint64_t) k * blockDim.x) * G, outputs[k]); }
			#pragma unroll
			for (unsigned k = 0; k < K; k++) { store_group<Stores>(out + (first_group + (uint64_t) k * blockDim.x) * G, outputs[k]); }
			continue;
		}
		#pragma unroll
		for (unsigned k = 0; k < K; k++) {
			uint64_t g = first_group + (uint64_t) k * blockDim.x;
			if (g >= num_groups) { continue; }
			group<Out, G> outputs;
			produce_group(g * G, outputs);
			store_group<Stores>(out + g * G, outputs);
		}
	}
	const uint64_t global_thread = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t num_threads = (uint64_t) gridDim.x * blockDim.x;
	for (uint64_t p = num_groups * G + global_thread; p < n; p += num_threads) { out[p] = produce_one(p); }
}

} // namespace stream_tiles
} // namespace cuda
