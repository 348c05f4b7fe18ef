// cuda::array<T,N> — trivially copyable fixed-size array usable on host and device; the type of
// model coefficient records (reference: src/cuda/stl_array.cuh:49-77).
#pragma once
#include <cstddef>

#if defined(__HIPCC__)
#define GIDDY_HD __host__ __device__ inline __attribute__((always_inline))
#else
#define GIDDY_HD inline
#endif

namespace cuda {

template <typename T, size_t N>
struct array {
	using value_type = T;
	using size_type = size_t;
	T elements_[N > 0 ? N : 1];

	GIDDY_HD T& operator[](size_t i) { return elements_[i]; }
	GIDDY_HD const T& operator[](size_t i) const { return elements_[i]; }
	GIDDY_HD static constexpr size_t size() { return N; }
	GIDDY_HD T* data() { return elements_; }
	GIDDY_HD const T* data() const { return elements_; }
	GIDDY_HD T* begin() { return elements_; }
	GIDDY_HD const T* begin() const { return elements_; }
	GIDDY_HD T* end() { return elements_ + N; }
	GIDDY_HD const T* end() const { return elements_ + N; }
	GIDDY_HD void fill(const T& v) { for (size_t i = 0; i < N; i++) { elements_[i] = v; } }
};

} // namespace cuda
