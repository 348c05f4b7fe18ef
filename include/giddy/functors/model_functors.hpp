// Parametric sequence models used by Frame-of-Reference and Model decompression:
//   cuda::functors::unary::parametric_model::{constant, linear, zero}<IndexSize, T>, polynomial<IndexSize, T, N>
// Interface (index_type, coefficients_type, operator()) and value semantics follow the reference's
// src/cuda/model_functors.hpp:25-125:  constant -> c[0];  linear -> c[0] + c[1]*x with x an
// uint_t<IndexSize> (so the arithmetic is unsigned and wraps) converted to T on return.
// `polynomial` keeps the reference's value EXACTLY as written (model_functors.hpp:54-66) — which is not
// Σ c_k x^k: it starts from 1 and multiplies the power up before it is used, 1 + Σ_k c_k x^(k+1), all in T —
// because a caller who instantiates a kernel template with it (no wrapper is registered for it, in the reference
// or here) gets that column from the reference.  `linear` and `constant` are stand-alone types (their names, not
// their bases, are what factory keys embed).
#pragma once
#include "giddy/array.hpp"
#include "giddy/util/integer.hpp"

namespace cuda {
namespace functors {
namespace unary {
namespace parametric_model {

template <unsigned IndexSize, typename T, unsigned ModelDimension>
struct function_t {
	enum { index_size = IndexSize };
	using index_type        = util::uint_t<IndexSize>;
	using size_type         = index_type;
	using coefficient_type  = T;
	using argument_type     = index_type;
	using result_type       = T;
	using coefficients_type = cuda::array<coefficient_type, ModelDimension>;
	enum : unsigned { model_dimension = ModelDimension };

	GIDDY_HD function_t(const coefficients_type& c) : model_coefficients(c) { }

	const coefficients_type model_coefficients;
};

template <unsigned IndexSize, typename T, unsigned NumModelCoefficients>
struct polynomial : public function_t<IndexSize, T, NumModelCoefficients> {
	using parent = function_t<IndexSize, T, NumModelCoefficients>;
	using typename parent::index_type;
	using typename parent::coefficients_type;
	using parent::model_coefficients;
	GIDDY_HD polynomial(const coefficients_type& c) : parent(c) { }
	GIDDY_HD T operator()(const index_type& x) const
	{
		T result = 1;
		T power_of_x = 1;
		#pragma unroll
		for (unsigned k = 0; k < NumModelCoefficients; k++) {
			power_of_x *= x;
			result += model_coefficients[k] * power_of_x;
		}
		return result;
	}
};

template <unsigned IndexSize, typename T>
struct linear : public function_t<IndexSize, T, 2> {
	using parent = function_t<IndexSize, T, 2>;
	using typename parent::index_type;
	using typename parent::coefficients_type;
	using parent::model_coefficients;
	GIDDY_HD linear(const coefficients_type& c) : parent(c) { }
	GIDDY_HD T operator()(const index_type& x) const { return model_coefficients[0] + model_coefficients[1] * x; }
};

template <unsigned IndexSize, typename T>
struct constant : public function_t<IndexSize, T, 1> {
	using parent = function_t<IndexSize, T, 1>;
	using typename parent::index_type;
	using typename parent::coefficients_type;
	using parent::model_coefficients;
	GIDDY_HD constant(const coefficients_type& c) : parent(c) { }
	GIDDY_HD T operator()(const index_type&) const { return model_coefficients[0]; }
};

template <unsigned IndexSize, typename T>
struct zero : public function_t<IndexSize, T, 0> {
	using parent = function_t<IndexSize, T, 0>;
	using typename parent::index_type;
	using typename parent::coefficients_type;
	GIDDY_HD zero(const coefficients_type& c) : parent(c) { }
	GIDDY_HD T operator()(const index_type&) const { return 0; }
};

} // namespace parametric_model
} // namespace unary
} // namespace functors
} // namespace cuda
