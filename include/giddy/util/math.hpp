// Host integer helpers used by the launch-configuration resolver (names follow the reference's
// src/util/math.hpp so that user code calling util::div_rounding_up etc. keeps compiling).
#pragma once
#include <cstddef>
#include <type_traits>

namespace util {

template <typename T, typename S>
constexpr inline typename std::common_type<T, S>::type div_rounding_up(const T& dividend, const S& divisor)
{
	return (dividend + divisor - 1) / divisor;
}
template <typename T, typename S>
constexpr inline typename std::common_type<T, S>::type round_down(const T& x, const S& y) { return x - x % y; }
template <typename T, typename S>
constexpr inline typename std::common_type<T, S>::type round_up(const T& x, const S& y)
{
	return div_rounding_up(x, y) * y;
}
template <typename T> constexpr inline bool is_power_of_2(T x) { return x > 0 && (x & (x - 1)) == 0; }
template <typename T> constexpr inline T round_down_to_power_of_2(T x)
{
	T p = 1;
	if (x == 0) { return 0; }
	while ((x >>= 1) != 0) { p <<= 1; }
	return p;
}
/// rounds x down to a multiple of the power of two `p`
template <typename T> constexpr inline T round_down_to_power_of_2(T x, T p) { return x & ~(p - 1); }
template <typename T> constexpr inline T gcd(T a, T b) { return b == 0 ? a : gcd(b, a % b); }
template <typename T> constexpr inline T lcm(T a, T b) { return (a / gcd(a, b)) * b; }

template <typename T> struct size_in_bits { enum : size_t { value = sizeof(T) * 8 }; };

} // namespace util
