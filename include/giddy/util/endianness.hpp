// util::endianness_t — same enumerators and values as the reference's src/util/endianness.h:9-32
// (the value 1 of `little` shows up inside DZB-fixed factory keys as "(util::endianness_t)1").
#pragma once
#include <endian.h>

namespace util {

enum class endianness_t : bool { big, big_endian = big, little, little_endian = little };

inline constexpr endianness_t compilation_target_endianness() {
#if __BYTE_ORDER == __LITTLE_ENDIAN
	return endianness_t::little;
#else
	return endianness_t::big;
#endif
}

} // namespace util
