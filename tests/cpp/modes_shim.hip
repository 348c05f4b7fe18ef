// The reference's three documented modes of use (README.md:94-126; examples/src/modes_of_use/*.cu) for the four schemes whose wrappers
// pick among several kernels — Delta, run-length, run-position, variable-width discard-zero-bytes — behind a plain C interface, so that
// tests/test_gpu_modes.py and tools/modes_bench.py can launch the SAME device-resident column through each of them and compare bits and
// rates with mode A (the factory's type-erased wrapper, which they reach through include/giddy.h):
//
//   mode 1 "B"   the wrapper instantiated; cuda::kernel::enqueue_launch(kernel, stream, config, typed parameters...)
//                (instantiation_of_wrapper.cu:22-38)
//   mode 2 "C"   the kernel template launched directly with a configuration from its launch_config_resolution_params_t
//                (direct_use_of_kernel.cu:17-36)
//   mode 3 "C+"  <scheme>::launch<...>(stream, config, typed parameters...): the one-line replacement for mode C's launch
//   mode 4 "B*"  cuda::kernel::enqueue_launch through a reference to the BASE type cuda::kernel_t (src/cuda/kernel_wrapper.cuh:135-152)
//
// resolve: modes 1, 3, 4 the wrapper's typed resolve_launch_configuration against get_device_function().attributes(); mode 2 the
// parameter struct without function attributes, as the reference's caller does.
#include <cstdint>
#include <cstdio>
#include <exception>

#include "kernel_wrappers/decompression/delta.cu"
#include "kernel_wrappers/decompression/discard_zero_bytes_variable.cu"
#include "kernel_wrappers/decompression/run_length_encoding.cu"
#include "kernel_wrappers/decompression/run_position_encoding.cu"
#include "kernels/resolve_launch_configuration.h"

namespace {

using cuda::launch_configuration_t;
namespace ks = cuda::kernels::decompression;

thread_local char last_error[512] = "";

template <typename F>
int guarded(F&& f)
{
	try { f(); return 0; }
	catch (const std::exception& e) { std::snprintf(last_error, sizeof last_error, "%s", e.what()); return -1; }
	catch (...) { std::snprintf(last_error, sizeof last_error, "unknown exception"); return -1; }
}

void put(uint32_t* out, const launch_configuration_t& c)
{
	out[0] = c.grid_dimensions.x; out[1] = c.grid_dimensions.y; out[2] = c.grid_dimensions.z;
	out[3] = c.block_dimensions.x; out[4] = c.block_dimensions.y; out[5] = c.block_dimensions.z;
	out[6] = c.dynamic_shared_memory_size;
}

launch_configuration_t get(const uint32_t* in)
{
	return launch_configuration_t(cuda::dimensions_t(in[0], in[1], in[2]), cuda::dimensions_t(in[3], in[4], in[5]), in[6]);
}

const cuda::launch_configuration_limits_t no_limits { cuda::nullopt, cuda::nullopt, cuda::nullopt };

template <typename Kernel, typename Params, typename... Sizes>
launch_configuration_t resolve(int mode, Sizes... sizes)
{
	const auto properties = cuda::device::current::get().properties();
	if (mode == 2) { Params params(properties, sizes...); return cuda::kernels::resolve_launch_configuration(params); }
	Kernel kernel;
	return kernel.resolve_launch_configuration(properties, kernel.get_device_function().attributes(), sizes..., no_limits);
}

template <typename Kernel, typename Function, typename Free, typename... Parameters>
void launch(int mode, hipStream_t stream, const launch_configuration_t& config, Function direct, Free free_launch, Parameters... parameters)
{
	Kernel kernel;
	switch (mode) {
	case 1: cuda::kernel::enqueue_launch(kernel, stream, config, parameters...); break;
	case 2: cuda::enqueue_launch(direct, stream, config, parameters...); break;
	case 3: free_launch(stream, config, parameters...); break;
	case 4: { const cuda::kernel_t& base = kernel; cuda::kernel::enqueue_launch(base, stream, config, parameters...); break; }
	default: throw std::invalid_argument("mode 1 (B), 2 (C), 3 (C+) or 4 (B through the base type)");
	}
}

} // namespace

extern "C" {

const char* modes_last_error() { return last_error; }

// ---- Delta <4, 8, 2> and <4, 4, 1> ------------------------------------------------------------------------------------------------
#define MODES_DELTA(NAME, I, U, C) \
int modes_resolve_##NAME(int mode, uint64_t length, uint64_t segment_length, uint32_t* config) \
{ \
	return guarded([&] { put(config, resolve<ks::delta::kernel_t<I, U, C>, ks::delta::launch_config_resolution_params_t<I, U, C>>(mode, (size_t) length, (size_t) segment_length)); }); \
} \
int modes_launch_##NAME(int mode, void* stream, const uint32_t* config, void* decompressed, const void* compressed_input, const void* anchor_values, uint64_t length, uint64_t segment_length) \
{ \
	using size_type = cuda::size_type_by_index_size<I>; \
	return guarded([&] { \
		launch<ks::delta::kernel_t<I, U, C>>(mode, static_cast<hipStream_t>(stream), get(config), ks::delta::decompress<I, U, C>, \
			[](auto... a) { ks::delta::launch<I, U, C>(a...); }, \
			static_cast<cuda::uint_t<U>*>(decompressed), static_cast<const cuda::uint_t<C>*>(compressed_input), static_cast<const cuda::uint_t<U>*>(anchor_values), \
			(size_type) length, (size_type) segment_length); \
	}); \
}
MODES_DELTA(delta_4_8_2, 4, 8, 2)
MODES_DELTA(delta_4_4_1, 4, 4, 1)

// ---- run-length encoding <4, 4, 2> ----------------------------------------------------------------------------------------------------
int modes_resolve_rle_4_4_2(int mode, uint64_t length, uint64_t segment_length, uint32_t* config)
{
	return guarded([&] { put(config, resolve<ks::run_length_encoding::kernel_t<4, 4, 2>, ks::run_length_encoding::launch_config_resolution_params_t<4, 4, 2>>(mode, (size_t) length, (size_t) segment_length)); });
}
int modes_launch_rle_4_4_2(int mode, void* stream, const uint32_t* config, void* decompressed, const void* run_data, const void* run_lengths,
	const void* position_anchors, const void* intra_run_anchor_offsets, uint64_t segment_length, uint64_t num_segments, uint64_t num_element_runs, uint64_t length)
{
	return guarded([&] {
		launch<ks::run_length_encoding::kernel_t<4, 4, 2>>(mode, static_cast<hipStream_t>(stream), get(config), ks::run_length_encoding::decompress<4, 4, 2>,
			[](auto... a) { ks::run_length_encoding::launch<4, 4, 2>(a...); },
			static_cast<uint32_t*>(decompressed), static_cast<const uint32_t*>(run_data), static_cast<const uint16_t*>(run_lengths),
			static_cast<const uint32_t*>(position_anchors), static_cast<const uint16_t*>(intra_run_anchor_offsets),
			(uint32_t) segment_length, (uint32_t) num_segments, (uint32_t) num_element_runs, (uint32_t) length);
	});
}

// ---- run-position encoding <4, 4, 2, true> and <4, 4, 4, false> ---------------------------------------------------------------------------
#define MODES_RPE(NAME, P, RELATIVE) \
int modes_resolve_##NAME(int mode, uint64_t length, uint64_t segment_length, uint32_t* config) \
{ \
	return guarded([&] { put(config, resolve<ks::run_position_encoding::kernel_t<4, 4, P, RELATIVE>, ks::run_position_encoding::launch_config_resolution_params_t<4, 4, P, RELATIVE>>(mode, (size_t) length, (size_t) segment_length)); }); \
} \
int modes_launch_##NAME(int mode, void* stream, const uint32_t* config, void* decompressed, const void* run_data, const void* run_start_positions, \
	const void* position_anchors, uint64_t segment_length, uint64_t num_anchors, uint64_t num_element_runs, uint64_t length) \
{ \
	return guarded([&] { \
		launch<ks::run_position_encoding::kernel_t<4, 4, P, RELATIVE>>(mode, static_cast<hipStream_t>(stream), get(config), ks::run_position_encoding::decompress<4, 4, P, RELATIVE>, \
			[](auto... a) { ks::run_position_encoding::launch<4, 4, P, RELATIVE>(a...); }, \
			static_cast<uint32_t*>(decompressed), static_cast<const uint32_t*>(run_data), static_cast<const cuda::uint_t<P>*>(run_start_positions), \
			static_cast<const uint32_t*>(position_anchors), (uint32_t) segment_length, (uint32_t) num_anchors, (uint32_t) num_element_runs, (uint32_t) length); \
	}); \
}
MODES_RPE(rpe_4_4_2_relative, 2, true)
MODES_RPE(rpe_4_4_4_absolute, 4, false)

// ---- variable-width discard-zero-bytes <8, 8, 4> ------------------------------------------------------------------------------------------
int modes_resolve_dzbv_8_8_4(int mode, uint64_t length, uint64_t segment_length, uint32_t* config)
{
	namespace vw = ks::discard_zero_bytes::variable_width;
	return guarded([&] { put(config, resolve<vw::kernel_t<8, 8, 4>, vw::launch_config_resolution_params_t<8, 8, 4>>(mode, (size_t) length, (size_t) segment_length)); });
}
int modes_launch_dzbv_8_8_4(int mode, void* stream, const uint32_t* config, void* decompressed, const void* compressed_data, const void* packed_element_sizes,
	const void* position_anchors, uint64_t segment_length, uint64_t length, uint32_t min_represented_element_size, uint32_t bits_per_element_size)
{
	namespace vw = ks::discard_zero_bytes::variable_width;
	return guarded([&] {
		launch<vw::kernel_t<8, 8, 4>>(mode, static_cast<hipStream_t>(stream), get(config), vw::decompress<8, 8, 4>,
			[](auto... a) { vw::launch<8, 8, 4>(a...); },
			static_cast<uint64_t*>(decompressed), static_cast<const unsigned char*>(compressed_data), static_cast<const uint32_t*>(packed_element_sizes),
			static_cast<const uint64_t*>(position_anchors), (uint64_t) segment_length, (uint64_t) length, (unsigned) min_represented_element_size, (unsigned) bits_per_element_size);
	});
}

} // extern "C"
