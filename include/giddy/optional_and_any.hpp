// cuda::optional / cuda::any / cuda::any_cast / cuda::maybe_at.
// The reference aliases Boost types here (src/cuda/optional_and_any.hpp:14-44,62-98); Boost is not
// available in this toolchain and C++17 has the same vocabulary in the standard library, so the
// aliases point at std::.  A failed cast throws std::bad_any_cast where the reference throws
// boost::bad_any_cast; a missing key still throws std::out_of_range from map::at.
#pragma once
#include <any>
#include <optional>
#include <string>
#include <unordered_map>

namespace cuda {

constexpr std::nullopt_t nullopt = std::nullopt;
template <typename T> using optional = std::optional<T>;
using any = std::any;
using bad_any_cast = std::bad_any_cast;

template <typename T>
inline T any_cast(const any& operand) { return std::any_cast<T>(operand); }

template <typename T>
inline optional<T> any_cast(const optional<any>& operand)
{
	return operand ? optional<T>(std::any_cast<T>(operand.value())) : optional<T>(nullopt);
}

template <typename Map>
inline optional<typename Map::mapped_type> maybe_at(const Map& map, const typename Map::key_type& key)
{
	auto it = map.find(key);
	return it == map.end() ? optional<typename Map::mapped_type>(nullopt) : optional<typename Map::mapped_type>(it->second);
}

} // namespace cuda
