// What every kernel-wrapper header needs (reference: src/kernel_wrappers/common.h:5-45).
#pragma once
#include "giddy/common.hpp"
#include "giddy/kernel_wrappers/registered_wrapper.hpp"
#include "giddy/kernels/resolve_launch_configuration.hpp"
#include "giddy/util/endianness.hpp"
#include "giddy/util/math.hpp"
#include "giddy/util/static_block.hpp"

namespace cuda {

/// the value stored under `name`, which must hold exactly a T (std::out_of_range / bad_any_cast otherwise)
template <typename T>
inline T argument(const kernel_t::arguments_type& arguments, const char* name) { return cuda::any_cast<T>(arguments.at(name)); }

/// the value stored under `name` if there is one
template <typename T>
inline optional<T> optional_argument(const kernel_t::arguments_type& arguments, const char* name) { return cuda::any_cast<T>(cuda::maybe_at(arguments, name)); }

} // namespace cuda

/// util::endianness_t <-> cuda::endianness_t (kernel_wrappers/common.h:38-43)
inline constexpr cuda::endianness_t translate(util::endianness_t endianness)
{
	return endianness == util::endianness_t::little_endian ? cuda::endianness_t::little_endian : cuda::endianness_t::big_endian;
}
