// cuda::kernel_t — the abstract, stateless kernel wrapper; cuda::launch_configuration_limits_t;
// cuda::kernel::enqueue_launch / resolve_launch_configuration free functions; MAKE_KERNEL_ARGUMENTS.
// Interface follows the reference's src/cuda/kernel_wrapper.cuh:20-33 (limits), :53-128 (kernel_t),
// :135-208 (free functions), :214-240 (macro).
#pragma once
#include <cstring>
#include <stdexcept>
#include <string>
#include <tuple>
#include <type_traits>
#include <unordered_map>
#include <utility>

#include "giddy/api/device.hpp"
#include "giddy/api/types.hpp"
#include "giddy/batch.hpp"
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

	/// enqueue_launch() for a caller that keeps its map (a loop over a column's chunks updating a few entries in place): nothing is copied
	virtual void enqueue_launch_with(
		stream::id_t stream_id, const launch_configuration_t& launch_config, const arguments_type& args) const
	{
		enqueue_launch(stream_id, launch_config, args);
	}

	virtual launch_configuration_t resolve_launch_configuration(
		device::properties_t device_properties, arguments_type extra_arguments,
		launch_configuration_limits_t limits = { nullopt, nullopt, nullopt }) const
	{
		return resolve_launch_configuration(device_properties, get_device_function().attributes(), extra_arguments, limits);
	}

	virtual launch_configuration_t resolve_launch_configuration(
		device::properties_t device_properties, device_function::attributes_t kernel_function_attributes,
		arguments_type extra_arguments, launch_configuration_limits_t limits = { nullopt, nullopt, nullopt }) const = 0;

	/// The kernel of the reference's signature — what a direct launch gets — carrying, for wrappers that pick among several builds of
	/// their decoder, those builds as alternates: attributes() states limits that hold for every kernel the resolved configuration may
	/// be launched with (api/types.hpp).
	virtual const device_function_t get_device_function() const = 0;

	/// cuda::kernel::enqueue_launch(kernel, [stream,] config, parameters...) for a caller that holds only this base type: parameters[i]
	/// points at the i-th value as passed, sizes[i] is its size.  A wrapper with a typed launch() of its own (below) re-types them and
	/// goes through it — so that this call, the typed free function and the argument map all reach the same kernels; this default is the
	/// reference's behaviour (src/cuda/kernel_wrapper.cuh:141-152): get_device_function() launched with the values as given, missing
	/// trailing parameters reading as zero.
	virtual void enqueue_launch_of_packed_parameters(stream::id_t stream_id, const launch_configuration_t& launch_config,
		void* const* parameters, const size_t* sizes, size_t count) const
	{
		(void) sizes;
		cuda::enqueue_launch_of_packed(get_device_function().ptr(), stream_id, launch_config, parameters, count);
	}

	// ---- batches: many small columns, ONE launch (giddy/batch.hpp; this library's addition to the reference's interface) ----------------

	/// whether the two functions below do anything but throw
	virtual bool launches_batches() const { return false; }

	/// One more piece of `table`: the column `piece_arguments` describe — exactly what enqueue_launch() would be given for it — with the workgroups of
	/// `piece_config`, the configuration resolve_launch_configuration() resolved for THAT column.  Same exceptions as enqueue_launch() for the map.
	virtual void add_to_batch(batch::table_t& table, const launch_configuration_t& piece_config, const arguments_type& piece_arguments) const
	{
		(void) table; (void) piece_config; (void) piece_arguments;
		throw std::logic_error("this kernel wrapper does not launch batches");
	}

	/// every piece of the tables decoded by one launch on `stream_id` (tables: batch::uploaded_t::view(), or wherever the caller keeps them on the device)
	virtual void enqueue_launch_of_batch(stream::id_t stream_id, const batch::on_device_t& tables) const
	{
		(void) stream_id; (void) tables;
		throw std::logic_error("this kernel wrapper does not launch batches");
	}

	virtual ~kernel_t() { }
};

namespace kernel {

namespace detail {

/// Kernel::launch(stream, config, parameters...) is callable: the wrapper has ONE typed entry point that picks among its kernels
template <typename Kernel, typename = void, typename... Parameters>
struct has_typed_launch_impl : std::false_type { };
template <typename Kernel, typename... Parameters>
struct has_typed_launch_impl<Kernel,
	decltype(Kernel::launch(std::declval<stream::id_t>(), std::declval<const launch_configuration_t&>(), std::declval<Parameters>()...), void()),
	Parameters...> : std::true_type { };
template <typename Kernel, typename... Parameters>
using has_typed_launch = has_typed_launch_impl<Kernel, void, Parameters...>;

template <typename... Typed, size_t... Is>
inline void launch_packed_impl(void (*launch)(stream::id_t, const launch_configuration_t&, Typed...), stream::id_t stream_id,
	const launch_configuration_t& launch_config, void* const* parameters, const size_t* sizes, size_t count, std::index_sequence<Is...>)
{
	if (count > sizeof...(Typed)) { throw std::invalid_argument("more launch parameters than the kernel takes"); }
	std::tuple<std::decay_t<Typed>...> values { };   // (missing trailing parameters: zero / nullptr, as a direct launch's spare slots)
	const bool sizes_agree = (... && (Is >= count || sizes[Is] == sizeof(std::decay_t<Typed>)));
	if (!sizes_agree) { throw std::invalid_argument("a launch parameter does not have the size of the kernel's parameter in its place"); }
	(..., (Is < count ? (void) std::memcpy(&std::get<Is>(values), parameters[Is], sizeof(std::decay_t<Typed>)) : (void) 0));
	launch(stream_id, launch_config, std::get<Is>(values)...);
}

/// the values behind `parameters`, re-typed as launch()'s own parameter types
template <typename... Typed>
inline void launch_packed(void (*launch)(stream::id_t, const launch_configuration_t&, Typed...), stream::id_t stream_id,
	const launch_configuration_t& launch_config, void* const* parameters, const size_t* sizes, size_t count)
{
	launch_packed_impl(launch, stream_id, launch_config, parameters, sizes, count, std::index_sequence_for<Typed...>());
}

} // namespace detail

// A caller that holds the base type (src/cuda/kernel_wrapper.cuh:135-152) ...
template <typename... KernelParameters>
inline void enqueue_launch(const kernel_t& kernel, stream::id_t stream_id,
	const launch_configuration_t& launch_config, KernelParameters... parameters)
{
	void* pointers[] = { nullptr, const_cast<void*>(reinterpret_cast<const void*>(&parameters))... };
	const size_t sizes[] = { 0, sizeof(KernelParameters)... };
	kernel.enqueue_launch_of_packed_parameters(stream_id, launch_config, pointers + 1, sizes + 1, sizeof...(KernelParameters));
}

// ... and one that instantiated the wrapper (the reference's "mode of use B", examples/src/modes_of_use/instantiation_of_wrapper.cu:22-38)
// — the better match for it: the wrapper's typed launch(), the very function its type-erased enqueue_launch() calls.
template <typename Kernel, typename... KernelParameters,
	typename = std::enable_if_t<std::is_base_of<kernel_t, Kernel>::value && detail::has_typed_launch<Kernel, KernelParameters...>::value>>
inline void enqueue_launch(const Kernel&, stream::id_t stream_id,
	const launch_configuration_t& launch_config, KernelParameters... parameters)
{
	Kernel::launch(stream_id, launch_config, parameters...);
}

template <typename Kernel, typename... KernelParameters,
	typename = std::enable_if_t<std::is_base_of<kernel_t, Kernel>::value && detail::has_typed_launch<Kernel, KernelParameters...>::value>>
inline void enqueue_launch(const Kernel&, const launch_configuration_t& launch_config, KernelParameters... parameters)
{
	Kernel::launch(stream::default_stream_id, launch_config, parameters...);
}

template <typename Kernel, typename... KernelParameters,
	typename = std::enable_if_t<std::is_base_of<kernel_t, Kernel>::value && detail::has_typed_launch<Kernel, KernelParameters...>::value>>
inline void enqueue_launch(const Kernel&, device::id_t device_id, stream::id_t stream_id,
	const launch_configuration_t& launch_config, KernelParameters... parameters)
{
	device::current::scoped_override_t<> using_device(device_id);
	Kernel::launch(stream_id, launch_config, parameters...);
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
