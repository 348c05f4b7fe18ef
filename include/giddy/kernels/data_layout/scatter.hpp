// Naive scatter ("patching"):   target[indices[i]] = data[i],  i < data_length,  distinct indices.
//
// Same kernel template name, parameter list and semantics as the reference's
// src/kernels/data_layout/scatter.cuh:26-46.  Patches are ~0.1 % of a column (SURVEY.md §8d,
// config C3), so the kernel is tiny next to the decoder it follows on the same stream; it reads
// indices and data with wide coalesced loads (4 patches per lane per step) and issues the
// unavoidable scattered element stores.
#pragma once
#include <hip/hip_runtime.h>

#include "giddy/common.hpp"
#include "giddy/kernels/launch_config_resolution_params.hpp"
#include "giddy/on_device/wave64.hpp"
#include "giddy/util/math.hpp"

namespace cuda {
namespace kernels {
namespace scatter {

static const serialization_factor_t DefaultSerializationFactor { 16 };

namespace detail { enum : unsigned { patches_per_lane = 4 }; }

template <unsigned OutputIndexSize, unsigned ElementSize, unsigned InputIndexSize,
	serialization_factor_t SerializationFactor = DefaultSerializationFactor>
__global__ void scatter(
	uint_t<ElementSize>*            __restrict__  target,
	const uint_t<ElementSize>*      __restrict__  data,
	const uint_t<OutputIndexSize>*  __restrict__  indices,
	size_type_by_index_size<InputIndexSize>       data_length)
{
	using namespace memory_access;
	using element_type = uint_t<ElementSize>;
	using index_type = uint_t<OutputIndexSize>;
	constexpr unsigned G = detail::patches_per_lane;
	const uint64_t n = data_length;
	const bool vectorizable = is_aligned(data, G * ElementSize >= 16 ? 16 : G * ElementSize)
		&& is_aligned(indices, G * OutputIndexSize >= 16 ? 16 : G * OutputIndexSize);
	const uint64_t num_groups = vectorizable ? n / G : 0;
	const uint64_t global_thread = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t num_threads = (uint64_t) gridDim.x * blockDim.x;

	for (uint64_t g = global_thread; g < num_groups; g += num_threads) {
		auto where = load_group<index_type, G>(indices + g * G);
		auto what = load_group<element_type, G>(data + g * G);
		#pragma unroll
		for (unsigned j = 0; j < G; j++) { target[where[j]] = what[j]; }
	}
	for (uint64_t i = num_groups * G + global_thread; i < n; i += num_threads) { target[indices[i]] = data[i]; }
}

/// The same for ASCENDING indices that may REPEAT: of equal indices the LAST one's datum is written (what a sequential scatter leaves, and what the fused
/// decompress-and-patch kernels promise — `scatter` above races on repeats).  An extension without a wrapper of its own: launched by the fused wrappers where the
/// plain decoder followed by a scatter is the faster way (kernel_wrappers/decompression/patched/decode_then_scatter.hpp), at grid stride under any configuration.
template <unsigned OutputIndexSize, unsigned ElementSize>
__global__ void scatter_ascending(
	uint_t<ElementSize>*            __restrict__  target,
	const uint_t<ElementSize>*      __restrict__  data,
	const uint_t<OutputIndexSize>*  __restrict__  indices,
	uint64_t                                      data_length)
{
	using namespace memory_access;
	using element_type = uint_t<ElementSize>;
	using index_type = uint_t<OutputIndexSize>;
	constexpr unsigned G = detail::patches_per_lane;
	const uint64_t n = data_length;
	const bool vectorizable = is_aligned(data, G * ElementSize >= 16 ? 16 : G * ElementSize)
		&& is_aligned(indices, G * OutputIndexSize >= 16 ? 16 : G * OutputIndexSize);
	const uint64_t num_groups = vectorizable ? n / G : 0;
	const uint64_t global_thread = (uint64_t) blockIdx.x * blockDim.x + threadIdx.x;
	const uint64_t num_threads = (uint64_t) gridDim.x * blockDim.x;

	// (a wavefront's lanes take consecutive groups; the trip count is the wavefront's, so that every lane is there for the shuffle)
	const unsigned lane = wave64::lane_id();
	for (uint64_t first_of_wave = global_thread - lane; first_of_wave < num_groups; first_of_wave += num_threads) {
		const uint64_t g = first_of_wave + lane;
		const bool active = g < num_groups;
		group<index_type, G> where;
		group<element_type, G> what;
		if (active) { where = load_group<index_type, G>(indices + g * G); what = load_group<element_type, G>(data + g * G); }
		else { where[0] = 0; }
		// the index behind the group: the next lane's first — out of its register; the wavefront's last active lane asks memory
		const index_type neighbours = sizeof(index_type) <= 4 ? (index_type) __shfl_down((unsigned) where[0], 1, 64) : (index_type) __shfl_down((unsigned long long) where[0], 1, 64);
		if (!active) { continue; }
		const bool has_next = (g + 1) * G < n;
		const bool neighbour_has_it = lane != wave64::size - 1 && g + 1 < num_groups;
		const index_type next = neighbour_has_it ? neighbours : (has_next ? indices[(g + 1) * G] : (index_type) 0);
		bool repeated[G];
		#pragma unroll
		for (unsigned j = 0; j < G; j++) { repeated[j] = j + 1 < G ? where[j + 1] == where[j] : (has_next && next == where[j]); }
		// The stores are UNCONDITIONAL wherever the lane has one patch that is not repeated: a repeated one is sent to that patch's place with that patch's datum (the same
		// lane storing the same thing twice) — a store inside a branch makes the compiler wait for every earlier store before it (vmcnt(0): the count is unknown behind a
		// branch), and four such waits a lane cost this kernel 25 % against `scatter` (profiles/r06_fused_regret.txt).  A lane whose patches ALL repeat — a run of equal
		// positions longer than the lane — is rare: then the wavefront takes the branches.
		unsigned safe = G;
		#pragma unroll
		for (unsigned j = 0; j < G; j++) { if (!repeated[j]) { safe = j; } }
		if (wave64::ballot(safe == G) == 0) {
			index_type safe_where = where[0]; element_type safe_what = what[0];
			#pragma unroll
			for (unsigned j = 1; j < G; j++) { if (safe == j) { safe_where = where[j]; safe_what = what[j]; } }
			#pragma unroll
			for (unsigned j = 0; j < G; j++) { target[repeated[j] ? safe_where : where[j]] = repeated[j] ? safe_what : what[j]; }
		}
		else {
			#pragma unroll
			for (unsigned j = 0; j < G; j++) { if (!repeated[j]) { target[where[j]] = what[j]; } }
		}
	}
	for (uint64_t i = num_groups * G + global_thread; i < n; i += num_threads) {
		if (i + 1 < n && indices[i + 1] == indices[i]) { continue; }
		target[indices[i]] = data[i];
	}
}

template <unsigned OutputIndexSize, unsigned ElementSize, unsigned InputIndexSize,
	serialization_factor_t SerializationFactor = DefaultSerializationFactor>
class launch_config_resolution_params_t final : public kernels::launch_config_resolution_params_t {
public:
	using parent = kernels::launch_config_resolution_params_t;

	launch_config_resolution_params_t(
		device::properties_t device_properties_, device_function::attributes_t kernel_function_attributes_,
		size_t input_data_length_)
	: parent(device_properties_, kernel_function_attributes_, nullopt)
	{
		grid_construction_resolution  = thread;
		length                        = util::div_rounding_up(input_data_length_, (size_t) detail::patches_per_lane);
		serialization_option          = auto_maximized;
		quanta.threads_in_block       = warp_size;
		upper_limits.threads_in_block = 256;
		keep_gpu_busy_factor          = 1024;  // in practice: one tile per workgroup — the dispatcher hands tiles out in address order, which HBM likes best
		default_serialization_factor  = SerializationFactor;
	}

	launch_config_resolution_params_t(device::properties_t device_properties_, size_t input_data_length_)
	: launch_config_resolution_params_t(device_properties_,
		device_function_t(scatter<OutputIndexSize, ElementSize, InputIndexSize, SerializationFactor>).attributes(),
		input_data_length_) { }
};

} // namespace scatter
} // namespace kernels
} // namespace cuda
