// util::endianness_t and util::compilation_target_endianness().
// The enumerators and their values are API: the reference's src/util/endianness.h:9-32 defines
// big = 0 / little = 1, and the value 1 appears inside DZB-fixed factory keys as "(util::endianness_t)1".
#pragma once

namespace util {

enum class endianness_t : bool {
	big = false, big_endian = big,
	little = true, little_endian = little,
};

/// byte order of the machine this translation unit is compiled for (gfx950 and x86-64 are both little-endian)
constexpr endianness_t compilation_target_endianness()
{
	return __BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__ ? endianness_t::little : endianness_t::big;
}

} // namespace util
