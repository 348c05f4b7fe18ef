// util::type_name<T>() — demangled (libstdc++ / Itanium ABI) name of T.  Factory keys are built
// from it exactly as the reference does (src/util/type_name.hpp:44-70 feeding
// src/kernel_wrappers/registered_wrapper.h:67-74), so key strings match character for character:
// g++ 11, ROCm clang 22 and hipcc demangle to identical text (SURVEY.md §7 hard part 1).
#pragma once
#include <cxxabi.h>
#include <cstdlib>
#include <memory>
#include <string>
#include <typeinfo>
#include <type_traits>

namespace util {

template <typename T>
std::string type_name()
{
	using TR = typename std::remove_reference<T>::type;
	int status = 0;
	std::unique_ptr<char, void (*)(void*)> demangled(
		abi::__cxa_demangle(typeid(TR).name(), nullptr, nullptr, &status), std::free);
	return (demangled && status == 0) ? std::string(demangled.get()) : std::string(typeid(TR).name());
}

template <typename T> std::string type_name_of(const T&) { return type_name<T>(); }

} // namespace util
