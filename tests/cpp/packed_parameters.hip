// cuda::kernel::enqueue_launch through the BASE type (src/cuda/kernel_wrapper.cuh:135-152) re-types the caller's values for the wrapper's
// typed launch (include/giddy/kernel_wrapper.hpp): what is checked on the way — and, with --attributes, that a run decoder's
// get_device_function().attributes() are limits every build of the decoder accepts.  Built by __graft_entry__.build_cpp_test_tools(),
// run by tests/test_gpu_modes.py on the GPU.
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <vector>

#include "kernel_wrappers/decompression/delta.cu"
#include "kernel_wrappers/decompression/run_length_encoding.cu"
#include "kernel_wrappers/decompression/run_position_encoding.cu"
#include "kernel_wrappers/decompression/frame_of_reference.cu"
#include "cuda/model_functors.hpp"

static int failures = 0;
#define CHECK(condition) do { if (!(condition)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #condition); failures++; } } while (0)

namespace ks = cuda::kernels::decompression;

template <typename Wrapper>
static void check_attributes(const char* name)
{
	Wrapper wrapper;
	const auto function = wrapper.get_device_function();
	const auto all = function.attributes();
	CHECK(function.num_alternates() >= 2);
	for (unsigned i = 0; i <= function.num_alternates(); i++) {
		const void* kernel = i == 0 ? function.ptr() : function.alternate(i - 1);
		const auto one = cuda::device_function_t::attributes_of(kernel);
		std::printf("%s kernel %u: %d threads per block at most, %zu B static LDS, %d registers\n", name, i, one.maxThreadsPerBlock, (size_t) one.sharedSizeBytes, one.numRegs);
		// (builds the wrapper launches with a grid and block of its own — pre-pass and decoder of long segments in one kernel — are listed, but a configuration the
		// caller resolves is never used for them: their limits are not the caller's)
		if (i > function.num_alternates_with_callers_configuration()) { continue; }
		CHECK(all.maxThreadsPerBlock <= one.maxThreadsPerBlock);
		CHECK(all.sharedSizeBytes >= one.sharedSizeBytes);
		CHECK(all.numRegs >= one.numRegs);
	}
}

int main(int argc, char** argv)
{
	if (argc > 1 && std::strcmp(argv[1], "--attributes") == 0) {
		check_attributes<ks::run_length_encoding::kernel_t<4, 4, 2>>("run_length_encoding<4,4,2>");
		check_attributes<ks::run_length_encoding::kernel_t<4, 1, 1>>("run_length_encoding<4,1,1>");
		check_attributes<ks::run_length_encoding::kernel_t<8, 8, 8, 8>>("run_length_encoding<8,8,8,8>");
		check_attributes<ks::run_position_encoding::kernel_t<4, 4, 2, true>>("run_position_encoding<4,4,2,true>");
		check_attributes<ks::run_position_encoding::kernel_t<4, 4, 4, false>>("run_position_encoding<4,4,4,false>");
		check_attributes<ks::run_position_encoding::kernel_t<8, 4, 8, false, 8>>("run_position_encoding<8,4,8,false,8>");
		if (failures == 0) { std::printf("attributes OK\n"); }
		return failures != 0;
	}
	const size_t n = 1 << 16, L = 4096;
	std::vector<uint16_t> deltas(n, 3);
	std::vector<uint64_t> anchors(n / L), expect(n), got(n);
	for (size_t s = 0; s < n / L; s++) { anchors[s] = 1000 * s; }
	for (size_t i = 0; i < n; i++) { expect[i] = anchors[i / L] + 3 * (i % L + 1); }
	uint16_t* d_deltas; uint64_t* d_anchors; uint64_t* d_out;
	cuda::throw_if_error(hipMalloc(&d_deltas, n * 2), "hipMalloc"); cuda::throw_if_error(hipMalloc(&d_anchors, n / L * 8), "hipMalloc"); cuda::throw_if_error(hipMalloc(&d_out, n * 8), "hipMalloc");
	cuda::throw_if_error(hipMemcpy(d_deltas, deltas.data(), n * 2, hipMemcpyHostToDevice), "hipMemcpy");
	cuda::throw_if_error(hipMemcpy(d_anchors, anchors.data(), n / L * 8, hipMemcpyHostToDevice), "hipMemcpy");

	ks::delta::kernel_t<4, 8, 2> wrapper;
	const cuda::kernel_t& base = wrapper;
	const auto config = wrapper.resolve_launch_configuration(cuda::device::current::get().properties(), wrapper.get_device_function().attributes(), n, L,
		cuda::launch_configuration_limits_t { cuda::nullopt, cuda::nullopt, cuda::nullopt });
	const uint64_t* const_anchors = d_anchors; const uint16_t* const_deltas = d_deltas;

	// the right types through the base type: decodes
	cuda::kernel::enqueue_launch(base, config, d_out, const_deltas, const_anchors, (uint32_t) n, (uint32_t) L);
	cuda::throw_if_error(hipMemcpy(got.data(), d_out, n * 8, hipMemcpyDeviceToHost), "hipMemcpy");
	CHECK(got == expect);
	// a 64-bit length where the kernel takes 32 bits: refused (the reference would launch with misplaced bytes)
	bool refused = false;
	try { cuda::kernel::enqueue_launch(base, config, d_out, const_deltas, const_anchors, (uint64_t) n, (uint32_t) L); } catch (const std::invalid_argument&) { refused = true; }
	CHECK(refused);
	// more parameters than the kernel has: refused
	refused = false;
	try { cuda::kernel::enqueue_launch(base, config, d_out, const_deltas, const_anchors, (uint32_t) n, (uint32_t) L, (uint64_t*) nullptr, 1, 2); } catch (const std::invalid_argument&) { refused = true; }
	CHECK(refused);
	// the instantiated wrapper: implicit conversions as for any function call (size_t -> the kernel's 32-bit size type)
	std::fill(got.begin(), got.end(), 0);
	cuda::throw_if_error(hipMemset(d_out, 0, n * 8), "hipMemset");
	cuda::kernel::enqueue_launch(wrapper, config, d_out, const_deltas, const_anchors, n, L);
	cuda::throw_if_error(hipMemcpy(got.data(), d_out, n * 8, hipMemcpyDeviceToHost), "hipMemcpy");
	CHECK(got == expect);
	static_assert(cuda::kernel::detail::has_typed_launch<ks::delta::kernel_t<4, 8, 2>, uint64_t*, const uint16_t*, const uint64_t*, size_t, size_t>::value, "Delta has a typed launch");
	namespace models = cuda::functors::unary::parametric_model;
	static_assert(!cuda::kernel::detail::has_typed_launch<ks::frame_of_reference::kernel_t<4, int32_t, int16_t, models::constant<4, int32_t>>, int32_t*, const int16_t*, const void*, uint32_t, uint32_t>::value,
		"a wrapper of one kernel goes the reference's way: get_device_function() launched");
	if (failures == 0) { std::printf("packed_parameters OK\n"); }
	return failures != 0;
}
