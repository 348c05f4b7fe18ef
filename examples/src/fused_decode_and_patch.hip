// Beyond the reference's three modes of use: one launch that decompresses AND patches (this library's
// "...::patched::kernel_t<...>" extensions), through the same factory and argument maps.  The frame-of-reference
// known-answer test of the other examples, with five of its elements replaced — once with the patch positions as a
// plain ascending array, once in the compressed-indices form of the reference's compressed_indices_scatter (one run:
// baseline 5, 16-bit offsets), which the wrapper expands into a caller-provided scratch array on the way.
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <numeric>
#include <vector>

#include "cuda/api_wrappers.h"
#include "cuda/model_functors.hpp"
#include "kernel_wrappers/registered_wrapper.h"

using index_type        = uint32_t;
using uncompressed_type = int32_t;
using compressed_type   = int16_t;
using model_type = cuda::functors::unary::parametric_model::constant<sizeof(index_type), uncompressed_type>;
using model_coefficients_type = typename model_type::coefficients_type;
using wrapper = cuda::registered::kernel_t;

template <typename T>
static auto on_device(const std::vector<T>& host)
{
	auto buffer = cuda::memory::device::make_unique<T[]>(cuda::device::current::get(), host.size());
	cuda::memory::copy(buffer.get(), host.data(), host.size() * sizeof(T));
	return buffer;
}

int main()
{
	const char* const key =
		"decompression::frame_of_reference::patched::kernel_t<4u, int, short, "
		"cuda::functors::unary::parametric_model::constant<4u, int> >";
	if (!wrapper::canProduce(key)) { std::cerr << "the library does not register " << key << "\n"; return EXIT_FAILURE; }
	auto kernel = wrapper::produceSubclass(key);

	const index_type length = 2 << 12, segment_length = 2 << 10, num_segments = length / segment_length;
	std::vector<compressed_type> h_compressed(length);
	std::vector<model_coefficients_type> h_coefficients(num_segments);
	for (index_type s = 0; s < num_segments; s++) {
		h_coefficients[s][0] = (uncompressed_type) (100 * s);
		std::iota(h_compressed.begin() + s * segment_length, h_compressed.begin() + (s + 1) * segment_length, (compressed_type) 0);
	}
	const std::vector<index_type> h_positions = { 5, 17, 300, 2048, 8191 };
	const std::vector<uncompressed_type> h_values = { -1, -2, -3, -4, -5 };
	const index_type num_patches = (index_type) h_positions.size();

	auto d_decompressed = cuda::memory::device::make_unique<uncompressed_type[]>(cuda::device::current::get(), length);
	auto d_compressed = on_device(h_compressed);
	auto d_coefficients = on_device(h_coefficients);
	auto d_values = on_device(h_values);
	auto d_positions = on_device(h_positions);

	wrapper::arguments_type sizes;
	sizes["length"] = (size_t) length;
	sizes["segment_length"] = (size_t) segment_length;
	auto configuration = kernel->resolve_launch_configuration(
		cuda::device::current::get().properties(), kernel->get_device_function().attributes(), sizes);

	auto check = [&](const char* what) {
		cuda::device::current::get().synchronize();
		std::vector<uncompressed_type> h_decompressed(length);
		cuda::memory::copy(h_decompressed.data(), d_decompressed.get(), length * sizeof(uncompressed_type));
		std::vector<uncompressed_type> expected(length);
		for (index_type i = 0; i < length; i++) { expected[i] = (uncompressed_type) (100 * (i / segment_length) + i % segment_length); }
		for (index_type k = 0; k < num_patches; k++) { expected[h_positions[k]] = h_values[k]; }
		for (index_type i = 0; i < length; i++) {
			if (h_decompressed[i] != expected[i]) {
				std::cerr << what << ": unexpected value at index " << i << ": got " << h_decompressed[i] << " instead of " << expected[i] << ".\n";
				std::exit(EXIT_FAILURE);
			}
		}
	};

	// 1. plain ascending positions
	{
		wrapper::arguments_type arguments;
		arguments["decompressed"] = d_decompressed.get();
		arguments["compressed_input"] = (const compressed_type*) d_compressed.get();
		arguments["segment_model_coefficients"] = (const model_coefficients_type*) d_coefficients.get();
		arguments["patch_values"] = (const uncompressed_type*) d_values.get();
		arguments["patch_positions"] = (const index_type*) d_positions.get();
		arguments["length"] = length;
		arguments["segment_length"] = segment_length;
		arguments["num_patches"] = num_patches;
		kernel->enqueue_launch(cuda::stream::default_stream_id, configuration, arguments);
		check("plain positions");
	}

	// 2. the same positions compressed: one run of five patches, baseline 5, offsets of 16 bits
	{
		const std::vector<index_type> h_run_lengths = { num_patches }, h_baselines = { 5 }, h_offsets_start = { 0 }, h_anchors = { 0 };
		const std::vector<unsigned char> h_offset_sizes = { 16 };
		std::vector<uint16_t> h_offsets;
		for (auto position : h_positions) { h_offsets.push_back((uint16_t) (position - 5)); }
		h_offsets.resize(8);   // (pool padded to whole index_type's)
		auto d_run_lengths = on_device(h_run_lengths);
		auto d_baselines = on_device(h_baselines);
		auto d_offsets_start = on_device(h_offsets_start);
		auto d_anchors = on_device(h_anchors);
		auto d_offset_sizes = on_device(h_offset_sizes);
		auto d_offsets = on_device(h_offsets);
		auto d_scratch = cuda::memory::device::make_unique<index_type[]>(cuda::device::current::get(), num_patches);

		wrapper::arguments_type arguments;
		arguments["decompressed"] = d_decompressed.get();
		arguments["compressed_input"] = (const compressed_type*) d_compressed.get();
		arguments["segment_model_coefficients"] = (const model_coefficients_type*) d_coefficients.get();
		arguments["patch_values"] = (const uncompressed_type*) d_values.get();
		arguments["length"] = length;
		arguments["segment_length"] = segment_length;
		arguments["num_patches"] = num_patches;
		arguments["scatter_position_run_lengths"] = (const index_type*) d_run_lengths.get();
		arguments["scatter_position_run_individual_offset_sizes"] = (const unsigned char*) d_offset_sizes.get();
		arguments["scatter_position_run_baseline_values"] = (const index_type*) d_baselines.get();
		arguments["scatter_position_run_offsets_start_pos"] = (const index_type*) d_offsets_start.get();
		arguments["scatter_position_offset_bytes"] = reinterpret_cast<const unsigned char*>(d_offsets.get());
		arguments["scatter_position_anchors"] = (const index_type*) d_anchors.get();
		arguments["anchoring_period"] = (index_type) 64;
		arguments["num_scatter_position_runs"] = (index_type) 1;
		arguments["patch_positions_scratch"] = d_scratch.get();
		(void) hipMemset(d_decompressed.get(), 0, length * sizeof(uncompressed_type));
		kernel->enqueue_launch(cuda::stream::default_stream_id, configuration, arguments);
		check("compressed positions");
	}
	std::cout << "OK: frame of reference + 5 patches in one launch, positions plain and compressed\n";
	return EXIT_SUCCESS;
}
