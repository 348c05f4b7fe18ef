// HIP-native boundary types.
//
// The reference takes these from the cuda-api-wrappers submodule, which is an EMPTY directory in
// the reference tree (external/cuda-api-wrappers, .gitmodules:1-3).  What is defined here is the
// subset of that vocabulary the hot path's public API mentions — same identifiers and member
// names as used at src/kernel_wrappers/decompression/frame_of_reference.cu:94,
// src/kernels/resolve_launch_configuration.cpp:154,204,281-282 and
// src/kernels/launch_config_resolution_params.h:115-118 — implemented directly on the HIP runtime
// for CDNA4.  The namespace is spelled `cuda` because it is part of the API contract: factory keys
// embed "cuda::functors::unary::parametric_model::…" (examples/src/modes_of_use/
// factory_provided_type_erased_wrapper.cu:14-15).  It is an identifier, not a CUDA dependency.
#pragma once
#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>
#include <initializer_list>
#include <stdexcept>
#include <string>

namespace cuda {

using native_word_t          = unsigned;
using grid_dimension_t       = unsigned;
using grid_block_dimension_t = unsigned;
using shared_memory_size_t   = unsigned;        // LDS bytes; CDNA4 has 160 KiB per CU, so 16 bits are not enough
using serialization_factor_t = unsigned short;  // appears in DZB-fixed / scatter factory keys as "(unsigned short)32"
using standard_bit_container_t = unsigned;

// A wavefront on CDNA4 is 64 lanes; the reference's `warp_size` name is kept for the API,
// its value is the hardware's.
enum : native_word_t { warp_size = 64, log_warp_size = 6 };
enum : shared_memory_size_t { no_shared_memory = 0 };

enum class endianness_t : bool { big, big_endian = big, little, little_endian = little };
constexpr endianness_t compilation_target_endianness = endianness_t::little;

class runtime_error : public std::runtime_error {
public:
	runtime_error(hipError_t status, const std::string& what)
	: std::runtime_error(what + ": " + hipGetErrorString(status)), status_(status) { }
	hipError_t code() const { return status_; }
private:
	hipError_t status_;
};

inline void throw_if_error(hipError_t status, const char* what)
{
	if (status != hipSuccess) { throw runtime_error(status, what); }
}

struct dimensions_t {
	unsigned x { 1 }, y { 1 }, z { 1 };
	constexpr dimensions_t() = default;
	constexpr dimensions_t(unsigned x_, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) { }
	constexpr size_t volume() const { return (size_t) x * y * z; }
	constexpr bool operator==(const dimensions_t& o) const { return x == o.x && y == o.y && z == o.z; }
	constexpr bool operator==(unsigned n) const { return volume() == n; } // `grid_dimensions == 0`, compressed_indices_scatter.cu:69
	operator dim3() const { return dim3(x, y, z); }
};

struct launch_configuration_t {
	dimensions_t          grid_dimensions { 1 };
	dimensions_t          block_dimensions { 1 };
	shared_memory_size_t  dynamic_shared_memory_size { no_shared_memory };

	constexpr launch_configuration_t() = default;
	constexpr launch_configuration_t(dimensions_t grid, dimensions_t block, shared_memory_size_t smem = no_shared_memory)
	: grid_dimensions(grid), block_dimensions(block), dynamic_shared_memory_size(smem) { }
};

namespace linear_grid {
// 1-D launch configuration, as in src/cuda/linear_grid.h:17-39
struct launch_configuration_t {
	grid_dimension_t        grid_length { 1 };
	grid_block_dimension_t  block_length { 1 };
	shared_memory_size_t    dynamic_shared_memory_size { no_shared_memory };
	operator cuda::launch_configuration_t() const { return { grid_length, block_length, dynamic_shared_memory_size }; }
};
} // namespace linear_grid

namespace stream {
using id_t = hipStream_t;
const id_t default_stream_id = nullptr;
} // namespace stream

namespace device {

using id_t = int;

struct compute_capability_t {
	int major_, minor_;
	// The resolver's keep-the-GPU-busy rule multiplies processors by "warp schedulings per cycle"
	// (resolve_launch_configuration.cpp:281-282).  On CDNA4 that is the 4 SIMDs of a CU.
	constexpr unsigned max_warp_schedulings_per_processor_cycle() const { return 4; }
	constexpr unsigned max_resident_warps_per_processor() const { return 32; }
};

struct properties_t : public hipDeviceProp_t {
	properties_t() : hipDeviceProp_t{} { }
	properties_t(const hipDeviceProp_t& p) : hipDeviceProp_t(p) { }
	grid_block_dimension_t max_threads_per_block() const { return (grid_block_dimension_t) maxThreadsPerBlock; }
	compute_capability_t compute_capability() const { return { major, minor }; }

	// Nominal MI355X (gfx950) properties, for resolving launch configurations where no
	// device is present (CPU-only test containers).  Values: MI355X_MICROARCH.md chip table.
	static properties_t nominal_mi355x()
	{
		properties_t p;
		p.multiProcessorCount = 256;
		p.maxThreadsPerBlock = 1024;
		p.maxThreadsDim[0] = 1024; p.maxThreadsDim[1] = 1024; p.maxThreadsDim[2] = 1024;
		p.maxGridSize[0] = 2147483647; p.maxGridSize[1] = 65535; p.maxGridSize[2] = 65535;
		p.sharedMemPerBlock = 65536;
		p.maxSharedMemoryPerMultiProcessor = 163840;
		p.warpSize = 64;
		p.regsPerBlock = 131072;
		p.maxThreadsPerMultiProcessor = 2048;
		p.major = 9; p.minor = 5;
		p.clockRate = 2400000;
		p.totalGlobalMem = (size_t) 288 << 30;
		return p;
	}
};

} // namespace device

namespace device_function {

struct attributes_t : public hipFuncAttributes {
	attributes_t() : hipFuncAttributes{} { maxThreadsPerBlock = 1024; }
	attributes_t(const hipFuncAttributes& a) : hipFuncAttributes(a) { }
};

inline shared_memory_size_t maximum_dynamic_shared_memory_per_block(
	const attributes_t& attributes, const device::properties_t& properties)
{
	size_t per_block = properties.sharedMemPerBlock ? properties.sharedMemPerBlock : 65536;
	size_t statically_used = attributes.sharedSizeBytes;
	return (shared_memory_size_t) (per_block > statically_used ? per_block - statically_used : 0);
}

} // namespace device_function

/// A kernel as the wrappers name it: the entry point of the reference's signature, plus — for a wrapper that launches one of several
/// builds of its decoder with the configuration the caller resolved (the run decoders: columns of short or of long runs) — those builds
/// as ALTERNATES.  attributes() then states the limits that hold for every one of them (the fewest threads per block any of them
/// accepts, the most static LDS and registers any of them uses), so that "attributes -> resolve_launch_configuration -> enqueue_launch"
/// (examples/src/modes_of_use/factory_provided_type_erased_wrapper.cu:20-31) resolves a configuration that whichever build runs can take.
class device_function_t {
public:
	enum : unsigned { max_alternates = 6 };
	device_function_t(const void* ptr) : ptr_(ptr) { }
	device_function_t(const void* ptr, std::initializer_list<const void*> alternates) : ptr_(ptr)
	{
		for (const void* a : alternates) { if (num_alternates_ < max_alternates && a != nullptr && a != ptr) { alternates_[num_alternates_++] = a; } }
	}
	/// ... and builds a wrapper launches with a grid and block of ITS OWN choosing (a pre-pass and a decoder in one kernel, a workgroup per segment): listed
	/// among the alternates — they may be what runs — but attributes() does not let them narrow the limits of a configuration they are never launched with
	/// (their static LDS would also shift what the resolution reserves to hold a kernel to so many workgroups per compute unit)
	device_function_t(const void* ptr, std::initializer_list<const void*> alternates, std::initializer_list<const void*> launched_with_their_own_configuration)
	: device_function_t(ptr, alternates)
	{
		num_with_callers_configuration_ = num_alternates_;
		for (const void* a : launched_with_their_own_configuration) { if (num_alternates_ < max_alternates && a != nullptr && a != ptr) { alternates_[num_alternates_++] = a; } }
	}
	template <typename... Params>
	device_function_t(void (*kernel)(Params...)) : ptr_(reinterpret_cast<const void*>(kernel)) { }
	const void* ptr() const { return ptr_; }
	operator const void*() const { return ptr_; }
	unsigned num_alternates() const { return num_alternates_; }
	/// how many of the alternates — the first so many — run with the configuration the CALLER resolved (the rest: with one of the wrapper's own)
	unsigned num_alternates_with_callers_configuration() const { return num_with_callers_configuration_ == ~0u ? num_alternates_ : num_with_callers_configuration_; }
	const void* alternate(unsigned i) const { return i < num_alternates_ ? alternates_[i] : nullptr; }
	static device_function::attributes_t attributes_of(const void* kernel)
	{
		hipFuncAttributes a;
		throw_if_error(hipFuncGetAttributes(&a, kernel), "hipFuncGetAttributes");
		return a;
	}
	device_function::attributes_t attributes() const
	{
		device_function::attributes_t all = attributes_of(ptr_);
		for (unsigned i = 0; i < (num_with_callers_configuration_ == ~0u ? num_alternates_ : num_with_callers_configuration_); i++) {
			const device_function::attributes_t a = attributes_of(alternates_[i]);
			all.maxThreadsPerBlock = a.maxThreadsPerBlock < all.maxThreadsPerBlock ? a.maxThreadsPerBlock : all.maxThreadsPerBlock;
			all.maxDynamicSharedSizeBytes = a.maxDynamicSharedSizeBytes < all.maxDynamicSharedSizeBytes ? a.maxDynamicSharedSizeBytes : all.maxDynamicSharedSizeBytes;
			all.sharedSizeBytes = a.sharedSizeBytes > all.sharedSizeBytes ? a.sharedSizeBytes : all.sharedSizeBytes;
			all.numRegs = a.numRegs > all.numRegs ? a.numRegs : all.numRegs;
			all.localSizeBytes = a.localSizeBytes > all.localSizeBytes ? a.localSizeBytes : all.localSizeBytes;
			all.constSizeBytes = a.constSizeBytes > all.constSizeBytes ? a.constSizeBytes : all.constSizeBytes;
		}
		return all;
	}
private:
	const void* ptr_;
	const void* alternates_[max_alternates] = { };
	unsigned num_alternates_ = 0;
	unsigned num_with_callers_configuration_ = ~0u;   // (~0: all of them)
};

} // namespace cuda
