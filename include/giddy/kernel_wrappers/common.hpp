// What every kernel-wrapper header needs (reference: src/kernel_wrappers/common.h:5-45).
#pragma once
#include "giddy/common.hpp"
#include "giddy/kernel_wrappers/registered_wrapper.hpp"
#include "giddy/kernels/resolve_launch_configuration.hpp"
#include "giddy/util/endianness.hpp"
#include "giddy/util/math.hpp"
#include "giddy/util/static_block.hpp"

#include <mutex>

namespace cuda {

/// the value stored under `name`, which must hold exactly a T (std::out_of_range / bad_any_cast otherwise)
template <typename T>
inline T argument(const kernel_t::arguments_type& arguments, const char* name) { return cuda::any_cast<T>(arguments.at(name)); }

/// the value stored under `name` if there is one
template <typename T>
inline optional<T> optional_argument(const kernel_t::arguments_type& arguments, const char* name) { return cuda::any_cast<T>(cuda::maybe_at(arguments, name)); }

/// the current device's properties, asked of the runtime once per device and process (not per host thread: the pipeline's worker threads are new every call) —
/// for the wrappers that resolve a configuration of their own at launch time (Delta's parts and chained tiles, the fused wrappers' decode + scatter)
inline const device::properties_t& current_device_properties()
{
	enum : int { most_devices = 64 };
	static device::properties_t cached[most_devices];
	static std::once_flag asked[most_devices];
	const int id = device::current::get().id();
	if (id < 0 || id >= most_devices) { static thread_local device::properties_t uncached; uncached = device::get(id).properties(); return uncached; }
	std::call_once(asked[id], [id] { cached[id] = device::get(id).properties(); });
	return cached[id];
}

} // namespace cuda

/// util::endianness_t <-> cuda::endianness_t (kernel_wrappers/common.h:38-43)
inline constexpr cuda::endianness_t translate(util::endianness_t endianness)
{
	return endianness == util::endianness_t::little_endian ? cuda::endianness_t::little_endian : cuda::endianness_t::big_endian;
}
