// The binary reduction operators and unary pre-transformations the scan / reduction kernels are
// parameterised by: cuda::functors::{plus, multiplies, minimum, maximum, bit_and, bit_or, bit_xor}
// and {identity, cast, absolute}.  Member names (first_argument_type, second_argument_type,
// result_type, argument_type, neutral_value()) are the ones the reference's kernels rely on
// (src/cuda/functors.hpp:99-210, :296-300, :371-375; used by src/kernels/reduction/scan.cuh:144-164).
#pragma once
#include <limits>
#include <type_traits>

#include "giddy/array.hpp"   // GIDDY_HD

namespace cuda {
namespace functors {

template <typename LHS, typename RHS, typename Result>
struct binary_function {
	using first_argument_type = LHS;
	using second_argument_type = RHS;
	using result_type = Result;
};

template <typename Argument, typename Result>
struct unary_function {
	using argument_type = Argument;
	using result_type = Result;
};

template <typename LHS, typename RHS = LHS, typename Result = LHS>
struct plus : binary_function<LHS, RHS, Result> {
	GIDDY_HD Result operator()(const LHS& x, const RHS& y) const { return (Result) (x + y); }
	GIDDY_HD static Result neutral_value() { return 0; }
};

template <typename LHS, typename RHS = LHS, typename Result = LHS>
struct multiplies : binary_function<LHS, RHS, Result> {
	GIDDY_HD Result operator()(const LHS& x, const RHS& y) const { return (Result) (x * y); }
	GIDDY_HD static Result neutral_value() { return 1; }
};

template <typename T>
struct minimum : binary_function<T, T, T> {
	GIDDY_HD T operator()(const T& x, const T& y) const { return x < y ? x : y; }
	GIDDY_HD static T neutral_value() { return std::numeric_limits<T>::max(); }
};

template <typename T>
struct maximum : binary_function<T, T, T> {
	GIDDY_HD T operator()(const T& x, const T& y) const { return x > y ? x : y; }
	GIDDY_HD static T neutral_value() { return std::numeric_limits<T>::lowest(); }
};

template <typename T>
struct bit_and : binary_function<T, T, T> {
	GIDDY_HD T operator()(const T& x, const T& y) const { return x & y; }
	GIDDY_HD static T neutral_value() { return (T) ~(T) 0; }
};

template <typename T>
struct bit_or : binary_function<T, T, T> {
	GIDDY_HD T operator()(const T& x, const T& y) const { return x | y; }
	GIDDY_HD static T neutral_value() { return 0; }
};

template <typename T>
struct bit_xor : binary_function<T, T, T> {
	GIDDY_HD T operator()(const T& x, const T& y) const { return x ^ y; }
	GIDDY_HD static T neutral_value() { return 0; }
};

template <typename T>
struct identity : unary_function<T, T> {
	GIDDY_HD T operator()(const T& x) const { return x; }
};

template <typename CastFrom, typename CastTo>
struct cast : unary_function<CastFrom, CastTo> {
	GIDDY_HD CastTo operator()(const CastFrom& x) const { return static_cast<CastTo>(x); }
};

template <typename T>
struct absolute : unary_function<T, T> {
	GIDDY_HD T operator()(const T& x) const { return x < 0 ? (T) -x : x; }
};

} // namespace functors
} // namespace cuda
