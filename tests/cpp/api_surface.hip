// Host-only exercise of the C++ API surface (no kernel is launched, no device is touched): every wrapper
// template's typed and type-erased launch-configuration resolution against nominal MI355X properties,
// the declarative resolver with hand-made parameters, its exception behaviour, the factory, the
// argument-map helpers.  Built and run by tests/test_cpp_api.py on CPU.
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "cuda/model_functors.hpp"
#include "kernel_wrappers/registered_wrapper.h"
#include "kernel_wrappers/decompression/delta.cu"
#include "kernel_wrappers/decompression/dictionary.cu"
#include "kernel_wrappers/decompression/discard_zero_bytes_fixed.cu"
#include "kernel_wrappers/decompression/discard_zero_bytes_variable.cu"
#include "kernel_wrappers/decompression/frame_of_reference.cu"
#include "kernel_wrappers/decompression/incidence_bitmaps.cu"
#include "kernel_wrappers/decompression/model.cu"
#include "kernel_wrappers/decompression/run_length_encoding.cu"
#include "kernel_wrappers/decompression/run_position_encoding.cu"
#include "kernel_wrappers/data_layout/scatter.cu"
#include "kernel_wrappers/data_layout/compressed_indices_scatter.cu"

static int failures = 0;
#define CHECK(condition) do { if (!(condition)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #condition); failures++; } } while (0)
#define CHECK_THROWS(expression, exception_type) do { bool thrown = false; try { (void) (expression); } catch (const exception_type&) { thrown = true; } catch (...) { } \
	if (!thrown) { std::printf("FAILED %s:%d: %s did not throw %s\n", __FILE__, __LINE__, #expression, #exception_type); failures++; } } while (0)

namespace {

// a hand-made parameter set, the way a user of the declarative mechanism would write one
struct my_params : cuda::kernels::launch_config_resolution_params_t {
	my_params(cuda::device::properties_t p) : launch_config_resolution_params_t(p, cuda::device_function::attributes_t()) { }
};

} // namespace

int main()
{
	using namespace cuda;
	namespace ks = cuda::kernels;
	namespace models = cuda::functors::unary::parametric_model;
	const auto props = device::properties_t::nominal_mi355x();
	const device_function::attributes_t attrs;
	const launch_configuration_limits_t no_limits { nullopt, nullopt, nullopt };

	// ---- typed resolution of every wrapper family --------------------------------------------------------
	{
		ks::decompression::frame_of_reference::kernel_t<4, int32_t, int16_t, models::constant<4, int32_t>> k;
		auto c = k.resolve_launch_configuration(props, attrs, (size_t) 1 << 20, (size_t) 4096);
		CHECK(c.block_dimensions.x == 256 && c.grid_dimensions.x == (1u << 20) / (256 * 8));   // (16-bit offsets to 32-bit values: two 16-byte stores per lane per tile)
		CHECK_THROWS(k.resolve_launch_configuration(props, attrs, (size_t) 100, (size_t) 0), std::invalid_argument);
		CHECK_THROWS(k.resolve_launch_configuration(props, attrs, (size_t) 100, (size_t) 6), std::invalid_argument);
	}
	{
		ks::decompression::dictionary::kernel_t<4, 4, 1> k;
		auto c = k.resolve_launch_configuration(props, attrs, (size_t) 256, (size_t) 1 << 30, nullopt, nullopt, no_limits);
		CHECK(c.dynamic_shared_memory_size == 26624 && c.grid_dimensions.x == 262144);   // (the 1 KiB dictionary, padded: 6 workgroups per CU)
		c = k.resolve_launch_configuration(props, attrs, (size_t) 256, (size_t) 1 << 30, optional<bool>(false), nullopt, no_limits);
		CHECK(c.dynamic_shared_memory_size == 0);
		c = k.resolve_launch_configuration(props, attrs, (size_t) 256, (size_t) 1 << 30, nullopt, optional<serialization_factor_t>(16), no_limits);
		CHECK(c.grid_dimensions.x == 262144);
	}
	{
		ks::decompression::delta::kernel_t<4, 8, 2> k;
		auto c = k.resolve_launch_configuration(props, attrs, (size_t) 1 << 30, (size_t) 4096, no_limits);
		CHECK(c.block_dimensions.x == 256 && c.grid_dimensions.x == 65536);
		CHECK_THROWS(k.resolve_launch_configuration(props, attrs, (size_t) 100, (size_t) 0, no_limits), std::invalid_argument);
	}
	{
		ks::decompression::discard_zero_bytes::fixed_width::kernel_t<4, 8, 2> k;
		CHECK(k.resolve_launch_configuration(props, attrs, (size_t) 1 << 20, no_limits).block_dimensions.x == 256);
		ks::decompression::discard_zero_bytes::variable_width::kernel_t<8, 8, 4> v;
		CHECK(v.resolve_launch_configuration(props, attrs, (size_t) 1 << 28, (size_t) 1024, no_limits).grid_dimensions.x == (1u << 18) / 4);
	}
	{
		ks::decompression::incidence_bitmaps::kernel_t<4, 4, 2, true> k;
		CHECK(k.resolve_launch_configuration(props, attrs, (size_t) 1 << 20, no_limits).grid_dimensions.x == (1u << 20) / 2048 / 4);
		ks::decompression::model::kernel_t<4, int32_t, models::linear<4, int32_t>> m;
		CHECK(m.resolve_launch_configuration(props, attrs, (size_t) 1 << 20, no_limits).grid_dimensions.x == (1u << 20) / (256 * 4));   // one 16-byte store per thread per tile
	}
	{
		ks::decompression::run_length_encoding::kernel_t<4, 4, 2> k;
		CHECK(k.resolve_launch_configuration(props, attrs, (size_t) 1 << 30, (size_t) 4096, no_limits).grid_dimensions.x == 65536);
		ks::decompression::run_position_encoding::kernel_t<4, 4, 2, true> p;
		CHECK(p.resolve_launch_configuration(props, attrs, (size_t) 1 << 30, (size_t) 4096, no_limits).grid_dimensions.x == 65536);
	}
	{
		ks::scatter::kernel_t<4, 8, 4> k;
		CHECK(k.resolve_launch_configuration(props, attrs, (size_t) 1000, no_limits).grid_dimensions.x == 1);
		ks::scatter::compressed_indices::kernel_t<4, 8, 4> c;
		CHECK(c.resolve_launch_configuration(props, attrs, (size_t) 5000, (size_t) 1024, no_limits).grid_dimensions.x == 2);
		CHECK_THROWS(c.resolve_launch_configuration(props, attrs, (size_t) 0, (size_t) 1024, no_limits), std::invalid_argument);
	}

	// ---- the declarative resolver, directly -----------------------------------------------------------------
	{
		my_params p(props);
		p.length = 1000; p.grid_construction_resolution = my_params::thread;
		p.serialization_option = my_params::fixed_factor; p.default_serialization_factor = 4;
		auto c = ks::resolve_launch_configuration(p);
		CHECK(c.block_dimensions.x == 250 && c.grid_dimensions.x == 1);                       // 1000 / 4 length units, one per thread
		CHECK_THROWS(ks::resolve_launch_configuration(p, no_limits, optional<serialization_factor_t>(8)), std::logic_error);
		p.serialization_option = my_params::none;
		CHECK_THROWS(ks::resolve_launch_configuration(p, no_limits, optional<serialization_factor_t>(2)), std::logic_error);
		p.quanta.threads_in_block = warp_size; p.length = 1 << 20;
		c = ks::resolve_launch_configuration(p);
		CHECK(c.block_dimensions.x == 1024 && c.grid_dimensions.x == 1024);
		CHECK_THROWS(ks::resolve_launch_configuration(p, launch_configuration_limits_t { nullopt, optional<grid_dimension_t>(10), nullopt }), std::invalid_argument);
		CHECK_THROWS(ks::resolve_launch_configuration(p, launch_configuration_limits_t::limit_block_size(32)), std::invalid_argument);
		p.grid_construction_resolution = my_params::block; p.block_resolution_constraints.fixed_threads_per_block = 2048; p.length = 5;
		CHECK_THROWS(ks::resolve_launch_configuration(p), std::invalid_argument);               // a fixed block length the device cannot honour
		p.block_resolution_constraints.fixed_threads_per_block = 192;
		c = ks::resolve_launch_configuration(p);
		CHECK(c.block_dimensions.x == 192 && c.grid_dimensions.x == 5);
		p.dynamic_shared_memory_requirement.per_length_unit = 16; p.dynamic_shared_memory_requirement.per_block = 100;
		CHECK(ks::resolve_launch_configuration(p).dynamic_shared_memory_size == 100 + 192 * 16);
		p.serialization_option = my_params::auto_maximized; p.keep_gpu_busy_factor = 2; p.length = 1 << 20;
		CHECK(ks::resolve_launch_configuration(p).grid_dimensions.x == 256 * 4 * 2 / 3);        // CUs x SIMDs x factor / wavefronts per block
	}

	// ---- factory + type-erased path -----------------------------------------------------------------------------
	{
		auto& factory = registered::kernel_t::getSubclassFactory();
		size_t reference_registrations = 0, fused_patch_extensions = 0, position_expansions = 0, compressors = 0;
		for (const auto& entry : factory.getInsantiators()) {
			(entry.first.rfind("compression::", 0) == 0 ? compressors :
			 entry.first.find("::patched::") != std::string::npos ? fused_patch_extensions :
			 entry.first.find("::positions::") != std::string::npos ? position_expansions : reference_registrations)++;
		}
		CHECK(reference_registrations == 163);
		CHECK(fused_patch_extensions == 12 + 20 + 8 + 5 + 12);
		CHECK(position_expansions == 8);
		CHECK(compressors == 112);
		const char* key = "decompression::frame_of_reference::kernel_t<4u, int, short, cuda::functors::unary::parametric_model::constant<4u, int> >";
		CHECK(factory.canProduce(key));
		auto k = registered::kernel_t::produceSubclass(key);
		size_t length = 1 << 20, segment_length = 4096;
		auto c = k->resolve_launch_configuration(props, attrs, MAKE_KERNEL_ARGUMENTS(length, segment_length));
		CHECK(c.block_dimensions.x == 256);
		CHECK_THROWS(registered::kernel_t::produceSubclass("no such kernel"), std::invalid_argument);
		CHECK_THROWS(k->resolve_launch_configuration(props, attrs, { { "length", length } }), std::out_of_range);
		CHECK_THROWS(k->resolve_launch_configuration(props, attrs, { { "length", (int) 5 }, { "segment_length", segment_length } }), cuda::bad_any_cast);
		CHECK(k->get_device_function().ptr() != nullptr);

		// the fused decompress-and-patch extensions: same factory, same resolver inputs as the plain wrapper, its launch shape for Delta
		auto patched = registered::kernel_t::produceSubclass("decompression::delta::patched::kernel_t<4u, 8u, 2u>");
		auto plain = registered::kernel_t::produceSubclass("decompression::delta::kernel_t<4u, 8u, 2u>");
		auto cp = patched->resolve_launch_configuration(props, attrs, MAKE_KERNEL_ARGUMENTS(length, segment_length));
		auto cq = plain->resolve_launch_configuration(props, attrs, MAKE_KERNEL_ARGUMENTS(length, segment_length));
		CHECK(cp.grid_dimensions.x == cq.grid_dimensions.x && cp.block_dimensions.x == cq.block_dimensions.x);
		CHECK(patched->get_device_function().ptr() != plain->get_device_function().ptr());
		size_t num_dictionary_entries = 256, data_length = 1 << 24;
		auto pd = registered::kernel_t::produceSubclass("decompression::dictionary::patched::kernel_t<4u, 4u, 1u>");
		auto cd = pd->resolve_launch_configuration(props, attrs, MAKE_KERNEL_ARGUMENTS(num_dictionary_entries, data_length));
		CHECK(cd.dynamic_shared_memory_size == 1024 && cd.block_dimensions.x == 256);
		CHECK(cd.grid_dimensions.x == (1u << 24) / (8 * 64 * 4) / 4);   // a wavefront per 8 rows of 64 x 4 elements, 4 wavefronts per workgroup
		CHECK_THROWS(pd->enqueue_launch(0, cd, { { "length", length } }), std::out_of_range);
	}

	// ---- cuda::any: what the reference's callers store, what the wrappers cast to ---------------------------------
	{
		int target[4] = { 1, 2, 3, 4 };
		int*       __restrict__ restricted = target;
		const int* __restrict__ restricted_to_const = target;
		int* const constant_pointer = target;
		const size_t constant_size = 7;
		any a = restricted, b = restricted_to_const, c = constant_pointer, d = constant_size;
		CHECK(any_cast<int*>(a) == target && a.type() == typeid(int*));          // top-level __restrict__ is dropped on the way in
		CHECK(any_cast<const int*>(b) == target);
		CHECK(any_cast<int*>(c) == target);                                      // as are const and volatile
		CHECK(any_cast<size_t>(d) == 7 && any_cast<const size_t&>(d) == 7);
		CHECK_THROWS(any_cast<const int*>(a), cuda::bad_any_cast);               // but pointee qualifiers are part of the type
		CHECK_THROWS(any_cast<int*>(b), cuda::bad_any_cast);
		CHECK_THROWS(any_cast<unsigned>(d), std::bad_any_cast);                  // (catchable as the standard exception too)
		CHECK_THROWS(any_cast<int>(any()), cuda::bad_any_cast);
		CHECK(any().empty() && !any().has_value() && any().type() == typeid(void) && d.has_value());
		any e = std::string("a value too big for the in-place buffer, so it lives on the heap .............");
		any f = e;                                                               // deep copy
		any g = std::move(e);
		CHECK(e.empty() && any_cast<std::string>(f) == any_cast<std::string>(g) && any_cast<std::string>(f).size() > 32);
		f = 5u; CHECK(any_cast<unsigned>(f) == 5u);
		f = d;  CHECK(any_cast<size_t>(f) == 7);
		f.swap(g); CHECK(any_cast<size_t>(g) == 7 && any_cast<std::string>(f).size() > 32);
		g.clear(); CHECK(g.empty());
		CHECK(any_cast<int>(&d) == nullptr && *any_cast<size_t>(&d) == 7);
		cuda::array<int, 1> record { { 42 } };
		any h = record; CHECK((any_cast<cuda::array<int, 1>>(h)[0] == 42));
		optional<any> none; CHECK(!any_cast<int>(none));
		// an argument map written the way factory_provided_type_erased_wrapper.cu:33-39 writes it
		size_t length = 16;
		kernel_t::arguments_type arguments = { { "decompressed", restricted }, { "compressed_input", restricted_to_const }, { "length", length } };
		CHECK(argument<int*>(arguments, "decompressed") == target && argument<const int*>(arguments, "compressed_input") == target);
		CHECK(!optional_argument<size_t>(arguments, "segment_length") && *optional_argument<size_t>(arguments, "length") == 16);
	}

	std::printf(failures == 0 ? "api_surface OK\n" : "api_surface: %d failure(s)\n", failures);
	return failures == 0 ? EXIT_SUCCESS : EXIT_FAILURE;
}
