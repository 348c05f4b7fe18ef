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

#include "giddy/kernels/tuning.hpp"
#include "giddy/on_device/wave64.hpp"

namespace cuda {
namespace stream_tiles {

enum : unsigned { steps_per_tile = GIDDY_STREAM_STEPS_PER_TILE };

template <typename Out> struct group_size { enum : unsigned { value = 16 / sizeof(Out) }; };

/// elements one thread produces per tile — what the launch parameters divide the column length by
template <typename Out> constexpr unsigned elements_per_thread_per_tile() { return group_size<Out>::value * steps_per_tile; }

/// out[p] = produce(p, in[p]) for p < n.  `produce_group(first_position, inputs, outputs)` handles G
/// consecutive elements at once (it may hoist per-group work such as a segment lookup).
template <typename In, typename Out, typename ProduceGroup, typename ProduceOne>
__device__ __forceinline__ void transform(
	Out* __restrict__ out, const In* __restrict__ in, uint64_t n, ProduceGroup produce_group, ProduceOne produce_one)
{
	using namespace memory_access;
	constexpr unsigned G = group_size<Out>::value;
	constexpr unsigned K = steps_per_tile;
	constexpr unsigned in_bytes = G * sizeof(In) >= 16 ? 16 : G * sizeof(In);

	const bool vectorizable = is_aligned(out, 16) && is_aligned(in, in_bytes);
	const uint64_t num_groups = vectorizable ? n / G : 0;
	const uint64_t groups_per_tile = (uint64_t) blockDim.x * K;

	for (uint64_t tile = blockIdx.x; tile * groups_per_tile < num_groups; tile += gridDim.x) {
		const uint64_t first_group = tile * groups_per_tile + threadIdx.x;
		group<In, G> inputs[K];
		if ((tile + 1) * groups_per_tile <= num_groups) {
			// a whole tile (every tile but possibly the last): no per-step bounds checks, so the K loads and the
			// K stores form straight-line code the compiler can schedule with counted waits
			#pragma unroll
			for (unsigned k = 0; k < K; k++) { inputs[k] = load_group<In, G>(in + (first_group + (uint64_t) k * blockDim.x) * G); }
			#pragma unroll
			for (unsigned k = 0; k < K; k++) {
				const uint64_t g = first_group + (uint64_t) k * blockDim.x;
				group<Out, G> outputs;
				produce_group(g * G, inputs[k], outputs);
				store_group(out + g * G, outputs);
			}
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
			produce_group(g * G, inputs[k], outputs);
			store_group(out + g * G, outputs);
		}
	}

	const uint64_t global_thread = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t num_threads = (uint64_t) gridDim.x * blockDim.x;
	for (uint64_t p = num_groups * G + global_thread; p < n; p += num_threads) { out[p] = produce_one(p, in[p]); }
}

/// out[p] = produce(p): the same loop for a decoder with no input array (Model)
template <typename Out, typename ProduceGroup, typename ProduceOne>
__device__ __forceinline__ void generate(Out* __restrict__ out, uint64_t n, ProduceGroup produce_group, ProduceOne produce_one)
{
	using namespace memory_access;
	constexpr unsigned G = group_size<Out>::value;
	constexpr unsigned K = steps_per_tile;
	const uint64_t num_groups = is_aligned(out, 16) ? n / G : 0;
	const uint64_t groups_per_tile = (uint64_t) blockDim.x * K;
	for (uint64_t tile = blockIdx.x; tile * groups_per_tile < num_groups; tile += gridDim.x) {
		const uint64_t first_group = tile * groups_per_tile + threadIdx.x;
		#pragma unroll
		for (unsigned k = 0; k < K; k++) {
			uint64_t g = first_group + (uint64_t) k * blockDim.x;
			if (g >= num_groups) { continue; }
			group<Out, G> outputs;
			produce_group(g * G, outputs);
			store_group(out + g * G, outputs);
		}
	}
	const uint64_t global_thread = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t num_threads = (uint64_t) gridDim.x * blockDim.x;
	for (uint64_t p = num_groups * G + global_thread; p < n; p += num_threads) { out[p] = produce_one(p); }
}

} // namespace stream_tiles
} // namespace cuda
