// What every kernel-wrapper header needs (reference: src/kernel_wrappers/common.h:5-45).
#pragma once
#include "giddy/common.hpp"
#include "giddy/kernel_wrappers/registered_wrapper.hpp"
#include "giddy/kernels/resolve_launch_configuration.hpp"
#include "giddy/util/endianness.hpp"
#include "giddy/util/math.hpp"
#include "giddy/util/static_block.hpp"

/// util::endianness_t <-> cuda::endianness_t (kernel_wrappers/common.h:38-43)
inline constexpr cuda::endianness_t translate(util::endianness_t endianness)
{
	return endianness == util::endianness_t::little_endian ? cuda::endianness_t::little_endian : cuda::endianness_t::big_endian;
}
