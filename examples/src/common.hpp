// Shared driver of the three "modes of use" examples: builds the frame-of-reference known-answer
// test the reference's examples use (length 8192, segments of 2048, segment s has reference value
// 100*s, offsets 0..2047 — examples/src/modes_of_use/common.cuh:31-49), runs the mode-specific
// decompress_on_device() and checks EVERY element (the reference's own check skips the last segment).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <iostream>
#include <numeric>
#include <vector>

#include "cuda/api_wrappers.h"
#include "cuda/model_functors.hpp"

using index_type        = uint32_t;
using size_type         = uint32_t;
using uncompressed_type = int32_t;
using compressed_type   = int16_t;
using model_type = cuda::functors::unary::parametric_model::constant<sizeof(index_type), uncompressed_type>;
using model_coefficients_type = typename model_type::coefficients_type;

void decompress_on_device(
	uncompressed_type*              decompressed,
	const compressed_type*          compressed,
	const model_coefficients_type*  segment_model_coefficients,
	size_type                       length,
	size_type                       segment_length);

int main()
{
	const size_t length = 2 << 12, segment_length = 2 << 10;
	const size_t num_segments = (length + segment_length - 1) / segment_length;

	std::vector<compressed_type> h_compressed(length);
	std::vector<model_coefficients_type> h_model_coefficients(num_segments);
	for (size_t s = 0; s < num_segments; s++) {
		h_model_coefficients[s][0] = (uncompressed_type) (100 * s);
		std::iota(h_compressed.begin() + s * segment_length, h_compressed.begin() + (s + 1) * segment_length, (compressed_type) 0);
	}

	auto device = cuda::device::current::get();
	auto d_decompressed       = cuda::memory::device::make_unique<uncompressed_type[]>(device, length);
	auto d_compressed         = cuda::memory::device::make_unique<compressed_type[]>(device, length);
	auto d_model_coefficients = cuda::memory::device::make_unique<model_coefficients_type[]>(device, num_segments);
	cuda::memory::copy(d_compressed.get(), h_compressed.data(), length * sizeof(compressed_type));
	cuda::memory::copy(d_model_coefficients.get(), h_model_coefficients.data(), num_segments * sizeof(model_coefficients_type));

	decompress_on_device(d_decompressed.get(), d_compressed.get(), d_model_coefficients.get(), (size_type) length, (size_type) segment_length);
	device.synchronize();

	std::vector<uncompressed_type> h_decompressed(length);
	cuda::memory::copy(h_decompressed.data(), d_decompressed.get(), length * sizeof(uncompressed_type));
	for (size_t s = 0; s < num_segments; s++) {
		for (size_t i = 0; i < segment_length; i++) {
			auto expected = (uncompressed_type) (100 * s + i);
			if (h_decompressed[s * segment_length + i] != expected) {
				std::cerr << "Unexpected value at index " << s * segment_length + i << " (segment " << s << ", offset " << i
					<< "): got " << h_decompressed[s * segment_length + i] << " instead of " << expected << ".\n";
				return EXIT_FAILURE;
			}
		}
	}
	std::cout << "OK: " << length << " elements decompressed correctly\n";
	return EXIT_SUCCESS;
}
