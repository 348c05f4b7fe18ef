// cuda::kernel_t — the abstract, stateless kernel wrapper; cuda::launch_configuration_limits_t;
// cuda::kernel::enqueue_launch / resolve_launch_configuration free functions; MAKE_KERNEL_ARGUMENTS.
// Interface follows the reference's src/cuda/kernel_wrapper.cuh:20-33 (limits), :53-128 (kernel_t),
// :135-208 (free functions), :214-240 (macro).
#pragma once
#include <string>
#include <unordered_map>

#include "giddy/api/device.hpp"
#include "giddy/api/types.hpp"
#include "giddy/optional_and_any.hpp"

namespace cuda {

struct launch_configuration_limits_t {
	optional<grid_block_dimension_t> block_size;
	optional<grid_dimension_t>       grid_size;
	optional<shared_memory_size_t>   dynamic_shared_memory;

	static launch_configuration_limits_t limit_block_size(optional<grid_block_dimension_t> limit) { return { limit, nullopt, nullopt }; }
	bool none() const { return !block_size && !grid_size && !dynamic_shared_memory; }
};

struct kernel_t {
	/// Type-erased arguments; values must hold EXACTLY the type the wrapper casts to.
	using arguments_type = std::unordered_map<std::string, any>;

	virtual void enqueue_launch(
		stream::id_t stream_id, const launch_configuration_t& launch_config, arguments_type args) const = 0;

	virtual launch_configuration_t resolve_launch_configuration(
		device::properties_t device_properties, arguments_type extra_arguments,
		launch_configuration_limits_t limits = { nullopt, nullopt, nullopt }) const
	{
		return resolve_launch_configuration(device_properties, get_device_function().attributes(), extra_arguments, limits);
	}

	virtual launch_configuration_t resolve_launch_configuration(
		device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
		arguments_type extra_arguments, launch_configuration_limits_t limits = { nullopt, nullopt, nullopt }) const = 0;

	virtual const device_function_t get_device_function() const = 0;

	virtual ~kernel_t() { }
};

namespace kernel {

template <typename... KernelParameters>
inline void enqueue_launch(const kernel_t& kernel, stream::id_t stream_id,
	const launch_configuration_t& launch_config, KernelParameters... parameters)
{
	cuda::enqueue_launch(kernel.get_device_function().ptr(), stream_id, launch_config, parameters...);
}

template <typename... KernelParameters>
inline void enqueue_launch(const kernel_t& kernel, const launch_configuration_t& launch_config, KernelParameters... parameters)
{
	enqueue_launch(kernel, stream::default_stream_id, launch_config, parameters...);
}

template <typename... KernelParameters>
inline void enqueue_launch(const kernel_t& kernel, device::id_t device_id, stream::id_t stream_id,
	const launch_configuration_t& launch_config, KernelParameters... parameters)
{
	device::current::scoped_override_t<> using_device(device_id);
	enqueue_launch(kernel, stream_id, launch_config, parameters...);
}

inline launch_configuration_t resolve_launch_configuration(const kernel_t& kernel,
	device::properties_t device_properties, kernel_t::arguments_type extra_arguments, launch_configuration_limits_t limits = {})
{
	return kernel.resolve_launch_configuration(device_properties, kernel.get_device_function().attributes(), extra_arguments, limits);
}

inline launch_configuration_t resolve_launch_configuration(const kernel_t& kernel,
	device::id_t device_id, kernel_t::arguments_type extra_arguments, launch_configuration_limits_t limits = {})
{
	return resolve_launch_configuration(kernel, device::get(device_id).properties(), extra_arguments, limits);
}

} // namespace kernel
} // namespace cuda

#define GIDDY_KA_1(a)                 { #a, a }
#define GIDDY_KA_2(a, ...)            { #a, a }, GIDDY_KA_1(__VA_ARGS__)
#define GIDDY_KA_3(a, ...)            { #a, a }, GIDDY_KA_2(__VA_ARGS__)
#define GIDDY_KA_4(a, ...)            { #a, a }, GIDDY_KA_3(__VA_ARGS__)
#define GIDDY_KA_5(a, ...)            { #a, a }, GIDDY_KA_4(__VA_ARGS__)
#define GIDDY_KA_6(a, ...)            { #a, a }, GIDDY_KA_5(__VA_ARGS__)
#define GIDDY_KA_7(a, ...)            { #a, a }, GIDDY_KA_6(__VA_ARGS__)
#define GIDDY_KA_8(a, ...)            { #a, a }, GIDDY_KA_7(__VA_ARGS__)
#define GIDDY_KA_9(a, ...)            { #a, a }, GIDDY_KA_8(__VA_ARGS__)
#define GIDDY_KA_10(a, ...)           { #a, a }, GIDDY_KA_9(__VA_ARGS__)
#define GIDDY_KA_11(a, ...)           { #a, a }, GIDDY_KA_10(__VA_ARGS__)
#define GIDDY_KA_12(a, ...)           { #a, a }, GIDDY_KA_11(__VA_ARGS__)
#define GIDDY_KA_PICK(_1,_2,_3,_4,_5,_6,_7,_8,_9,_10,_11,_12,N,...) N
/// { {"arg1", arg1}, {"arg2", arg2}, ... } from identifiers named like the wrapper's keys
#define MAKE_KERNEL_ARGUMENTS(...) { GIDDY_KA_PICK(__VA_ARGS__, GIDDY_KA_12, GIDDY_KA_11, GIDDY_KA_10, GIDDY_KA_9, \
	GIDDY_KA_8, GIDDY_KA_7, GIDDY_KA_6, GIDDY_KA_5, GIDDY_KA_4, GIDDY_KA_3, GIDDY_KA_2, GIDDY_KA_1)(__VA_ARGS__) }
