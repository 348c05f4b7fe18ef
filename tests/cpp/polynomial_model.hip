// The reference's generic `parametric_model::polynomial` functor (src/cuda/model_functors.hpp:45-74) through the two kernel
// templates that take a model — no wrapper is registered for it, so this is the "direct use of a kernel" mode — checked
// against the same functor evaluated on the host, and against the closed form 1 + sum_k c_k x^(k+1) in wrapping arithmetic.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cuda/api_wrappers.h"
#include "cuda/model_functors.hpp"
#include "kernels/resolve_launch_configuration.h"
#include "kernels/decompression/model.cuh"
#include "kernels/decompression/frame_of_reference.cuh"

int main()
{
	using model_type = cuda::functors::unary::parametric_model::polynomial<4, int32_t, 3>;
	using coefficients_type = model_type::coefficients_type;
	const uint32_t length = (1u << 20) + 77, segment_length = 4096, num_segments = (length + segment_length - 1) / segment_length;
	auto device = cuda::device::current::get();
	auto properties = device.properties();
	int failures = 0;

	// ---- model::decompress: out[p] = polynomial(p) ----
	{
		const coefficients_type c { { -7, 3, 11 } };
		auto out = cuda::memory::device::make_unique<int32_t[]>(device, length);
		namespace ns = cuda::kernels::decompression::model;
		ns::launch_config_resolution_params_t<4, int32_t, model_type> params(properties, length);
		auto config = cuda::kernels::resolve_launch_configuration(params);
		cuda::launch(ns::decompress<4, int32_t, model_type>, config, out.get(), c, length);
		std::vector<int32_t> host(length);
		cuda::memory::copy(host.data(), out.get(), length * sizeof(int32_t));
		model_type on_host(c);
		for (uint32_t p = 0; p < length; p++) {
			const uint32_t x = p;
			const int32_t closed = (int32_t) (1u + (uint32_t) c[0] * x + (uint32_t) c[1] * x * x + (uint32_t) c[2] * x * x * x);
			if (host[p] != on_host(p) || host[p] != closed) { if (failures++ < 5) { std::printf("model: position %u: %d, host functor %d, closed form %d\n", p, host[p], on_host(p), closed); } }
		}
	}
	// ---- frame_of_reference::decompress: out[p] = polynomial_s(p - s * L) + offset[p] ----
	{
		std::vector<coefficients_type> records(num_segments);
		std::vector<int16_t> offsets(length);
		for (uint32_t s = 0; s < num_segments; s++) { records[s] = { { (int32_t) (s * 13) - 500, (int32_t) (s % 7) - 3, (int32_t) (s % 3) } }; }
		for (uint32_t p = 0; p < length; p++) { offsets[p] = (int16_t) ((p * 2654435761u) >> 16); }
		auto out = cuda::memory::device::make_unique<int32_t[]>(device, length);
		auto in = cuda::memory::device::make_unique<int16_t[]>(device, length);
		auto coefficients = cuda::memory::device::make_unique<coefficients_type[]>(device, num_segments);
		cuda::memory::copy(in.get(), offsets.data(), length * sizeof(int16_t));
		cuda::memory::copy(coefficients.get(), records.data(), num_segments * sizeof(coefficients_type));
		namespace ns = cuda::kernels::decompression::frame_of_reference;
		ns::launch_config_resolution_params_t<4, int32_t, int16_t, model_type> params(properties, length, segment_length);
		auto config = cuda::kernels::resolve_launch_configuration(params);
		cuda::launch(ns::decompress<4, int32_t, int16_t, model_type>, config, out.get(), in.get(), coefficients.get(), length, segment_length, 1u);
		std::vector<int32_t> host(length);
		cuda::memory::copy(host.data(), out.get(), length * sizeof(int32_t));
		for (uint32_t p = 0; p < length; p++) {
			const int32_t expect = (int32_t) ((uint32_t) model_type(records[p / segment_length])(p % segment_length) + (uint32_t) (int32_t) offsets[p]);
			if (host[p] != expect) { if (failures++ < 5) { std::printf("frame of reference: position %u: %d instead of %d\n", p, host[p], expect); } }
		}
	}
	std::printf(failures == 0 ? "polynomial_model OK\n" : "polynomial_model: %d mismatches\n", failures);
	return failures == 0 ? EXIT_SUCCESS : EXIT_FAILURE;
}
