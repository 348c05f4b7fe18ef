// Names shared by kernels and wrappers: cuda::uint_t, promoted_size, size_type_by_index_size
// (reference: src/kernels/common.cuh:19-53, src/cuda/miscellany.h:10-31).
#pragma once
#include <type_traits>

#include "giddy/api/types.hpp"
#include "giddy/util/integer.hpp"

namespace cuda {

using util::uint_t;

/// Index variables narrower than a register are promoted to the 32-bit native word.
template <typename Size>
using promoted_size = typename std::common_type<Size, native_word_t>::type;

/// uint32_t for index sizes 1, 2, 4; uint64_t for 8.
template <unsigned N>
using size_type_by_index_size = promoted_size<uint_t<N>>;

template <typename T>
using size_type_by_index_type = size_type_by_index_size<sizeof(T)>;

} // namespace cuda
