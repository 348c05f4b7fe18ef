// Device handle, launch and memory helpers (HIP runtime) — the few cuda-api-wrappers entry
// points the reference's three "modes of use" call (examples/src/modes_of_use/common.cuh:78-89,
// direct_use_of_kernel.cu:16-36): device::current::get().properties(), cuda::launch,
// cuda::enqueue_launch, memory::device::make_unique, memory::copy.
#pragma once
#include <hip/hip_runtime.h>

#include <memory>
#include <type_traits>

#include "giddy/api/types.hpp"

namespace cuda {

namespace device {

class device_t {
public:
	explicit device_t(id_t id) : id_(id) { }
	id_t id() const { return id_; }
	properties_t properties() const
	{
		hipDeviceProp_t p;
		throw_if_error(hipGetDeviceProperties(&p, id_), "hipGetDeviceProperties");
		return p;
	}
	void synchronize() const
	{
		scoped_set s(id_);
		throw_if_error(hipDeviceSynchronize(), "hipDeviceSynchronize");
	}
private:
	struct scoped_set {
		int previous;
		explicit scoped_set(int id) { throw_if_error(hipGetDevice(&previous), "hipGetDevice"); if (id != previous) throw_if_error(hipSetDevice(id), "hipSetDevice"); }
		~scoped_set() { (void) hipSetDevice(previous); }
	};
	id_t id_;
};

inline device_t get(id_t id) { return device_t(id); }

namespace current {
inline device_t get()
{
	int id;
	throw_if_error(hipGetDevice(&id), "hipGetDevice");
	return device_t(id);
}
inline void set(id_t id) { throw_if_error(hipSetDevice(id), "hipSetDevice"); }

template <bool AssumedCurrent = false>
struct scoped_override_t {
	int previous;
	explicit scoped_override_t(id_t id) { throw_if_error(hipGetDevice(&previous), "hipGetDevice"); throw_if_error(hipSetDevice(id), "hipSetDevice"); }
	~scoped_override_t() { (void) hipSetDevice(previous); }
};
} // namespace current
} // namespace device

/// Enqueue `kernel_function` with `count` parameter values behind `parameters`.
/// The runtime reads one pointer per KERNEL parameter.  A caller may pass fewer values than the kernel has
/// parameters — the reference's own mode-B example gives frame_of_reference::decompress five of its six
/// (examples/src/modes_of_use/instantiation_of_wrapper.cu:31-38) — so spare slots follow the given ones, each
/// pointing at zeroes: the missing trailing parameters read as 0 / nullptr instead of whatever lies past the array.
inline void enqueue_launch_of_packed(const void* kernel_function, stream::id_t stream_id,
	const launch_configuration_t& launch_config, void* const* parameters, size_t count)
{
	if (launch_config.grid_dimensions.volume() == 0) { return; } // empty input: nothing to do
	enum : size_t { most = 24, spare_slots = 8 };
	if (count > most) { throw std::invalid_argument("more launch parameters than any kernel of this library takes"); }
	alignas(16) static const unsigned char zeroes[64] = { };
	void* argument_pointers[most + spare_slots];
	for (size_t i = 0; i < count; i++) { argument_pointers[i] = parameters[i]; }
	for (size_t i = count; i < count + spare_slots; i++) { argument_pointers[i] = const_cast<void*>(static_cast<const void*>(zeroes)); }
	throw_if_error(
		hipLaunchKernel(kernel_function, launch_config.grid_dimensions, launch_config.block_dimensions,
			argument_pointers, launch_config.dynamic_shared_memory_size, stream_id),
		"hipLaunchKernel");
}

/// Enqueue `kernel_function` on `stream_id` with the resolved configuration; arguments are passed
/// by value exactly as given (callers cast to the kernel's parameter types first).
template <typename KernelFunction, typename... KernelParameters>
inline void enqueue_launch(
	KernelFunction kernel_function, stream::id_t stream_id,
	const launch_configuration_t& launch_config, KernelParameters... parameters)
{
	void* pointers[] = { nullptr, const_cast<void*>(reinterpret_cast<const void*>(&parameters))... };
	enqueue_launch_of_packed(reinterpret_cast<const void*>(kernel_function), stream_id, launch_config, pointers + 1, sizeof...(KernelParameters));
}

template <typename KernelFunction, typename... KernelParameters>
inline void launch(KernelFunction kernel_function, const launch_configuration_t& launch_config, KernelParameters... parameters)
{
	enqueue_launch(kernel_function, stream::default_stream_id, launch_config, parameters...);
}

namespace memory {

inline void copy(void* destination, const void* source, size_t num_bytes)
{
	throw_if_error(hipMemcpy(destination, source, num_bytes, hipMemcpyDefault), "hipMemcpy");
}

namespace device {
struct deleter { void operator()(void* p) const { (void) hipFree(p); } };
template <typename T> using unique_ptr = std::unique_ptr<T, deleter>;

template <typename T>
inline unique_ptr<T> make_unique(const cuda::device::device_t& dev, size_t num_elements)
{
	static_assert(std::is_array<T>::value, "use make_unique<T[]>(device, n)");
	using element = typename std::remove_extent<T>::type;
	cuda::device::current::scoped_override_t<> use(dev.id());
	void* p = nullptr;
	throw_if_error(hipMalloc(&p, num_elements * sizeof(element)), "hipMalloc");
	return unique_ptr<T>(static_cast<element*>(p));
}
} // namespace device
} // namespace memory

} // namespace cuda
