// Sized integer aliases.  Mirrors the public names of the reference's src/util/integer.h:194-210
// (util::uint_t<N>, util::int_t<N,Signed>::type) for N in {1,2,4,8} — the only sizes any
// registered kernel instantiation uses (the reference's byte-array fallback for other N is
// "untested", integer.h:42, and never registered).
#pragma once
#include <cstdint>
#include <cstddef>
#include <type_traits>

namespace util {

namespace detail { enum : bool { is_unsigned = false, is_signed = true }; }

template <unsigned N, bool Signed = detail::is_signed> struct int_t;
template <> struct int_t<1, detail::is_signed  > { using type = int8_t;   };
template <> struct int_t<2, detail::is_signed  > { using type = int16_t;  };
template <> struct int_t<4, detail::is_signed  > { using type = int32_t;  };
template <> struct int_t<8, detail::is_signed  > { using type = int64_t;  };
template <> struct int_t<1, detail::is_unsigned> { using type = uint8_t;  };
template <> struct int_t<2, detail::is_unsigned> { using type = uint16_t; };
template <> struct int_t<4, detail::is_unsigned> { using type = uint32_t; };
template <> struct int_t<8, detail::is_unsigned> { using type = uint64_t; };

template <unsigned N> using uint_t = typename int_t<N, detail::is_unsigned>::type;
template <unsigned N> using sint_t = typename int_t<N, detail::is_signed>::type;

} // namespace util
