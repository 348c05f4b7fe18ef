// cuda::optional / cuda::any / cuda::any_cast / cuda::bad_any_cast / cuda::maybe_at.
//
// The reference aliases Boost types here (src/cuda/optional_and_any.hpp:14-44,62-98).  Boost is not in
// this toolchain; `optional` is the standard one, `any` is a small type-erased holder of its own because
// std::any does not accept what the reference's callers put into argument maps: pointers declared
// `T* __restrict__` (examples/src/modes_of_use/factory_provided_type_erased_wrapper.cu:8-13,33-39).
// boost::any takes them (the held type's top-level qualifiers do not take part in its typeid check);
// this one strips top-level const / volatile / __restrict__ when a value is stored, so that
// `any_cast<const short*>` finds what `{ "compressed_input", compressed }` stored whatever the qualifiers
// of the caller's variable.  A failed cast throws cuda::bad_any_cast — derived from std::bad_any_cast,
// where the reference throws boost::bad_any_cast; a missing key still throws std::out_of_range from map::at.
#pragma once
#include <any>
#include <cstddef>
#include <new>
#include <optional>
#include <string>
#include <type_traits>
#include <typeinfo>
#include <unordered_map>
#include <utility>

namespace cuda {

constexpr std::nullopt_t nullopt = std::nullopt;
template <typename T> using optional = std::optional<T>;

class bad_any_cast : public std::bad_any_cast {
public:
	const char* what() const noexcept override { return "bad any_cast"; }
};

namespace detail {
// the type an `any` holds for a value of type T: decayed, without top-level cv or __restrict__
template <typename T> struct stored { using type = std::remove_cv_t<__remove_restrict(std::decay_t<T>)>; };
template <typename T> using stored_t = typename stored<T>::type;
} // namespace detail

class any {
	enum class operation { copy, move, destroy, address, type };
	union storage_t {
		alignas(std::max_align_t) unsigned char in_place[2 * sizeof(void*)];
		void* on_heap;
	};
	using manager_t = const void* (*)(operation, const any* self, any* other);

	template <typename T>
	static constexpr bool fits = sizeof(T) <= sizeof(storage_t) && alignof(T) <= alignof(storage_t) && std::is_nothrow_move_constructible<T>::value;

	template <typename T>
	static const void* manage(operation op, const any* self, any* other)
	{
		auto held = [](const any* a) -> T* {
			if constexpr (fits<T>) { return reinterpret_cast<T*>(const_cast<unsigned char*>(a->storage.in_place)); }
			else { return static_cast<T*>(a->storage.on_heap); }
		};
		switch (op) {
		case operation::copy:
			if constexpr (fits<T>) { ::new (other->storage.in_place) T(*held(self)); }
			else { other->storage.on_heap = new T(*held(self)); }
			other->manager = self->manager;
			return nullptr;
		case operation::move: // self -> other; self is left empty
			if constexpr (fits<T>) { ::new (other->storage.in_place) T(std::move(*held(self))); held(self)->~T(); }
			else { other->storage.on_heap = self->storage.on_heap; }
			other->manager = self->manager;
			const_cast<any*>(self)->manager = nullptr;
			return nullptr;
		case operation::destroy:
			if constexpr (fits<T>) { held(self)->~T(); }
			else { delete held(self); }
			return nullptr;
		case operation::address: return held(self);
		case operation::type:    return &typeid(T);
		}
		return nullptr;
	}

	storage_t storage;
	manager_t manager { nullptr };

	template <typename T> friend const T* any_cast(const any*) noexcept;

public:
	constexpr any() noexcept : storage { } { }
	any(const any& other) : storage { } { if (other.manager) { other.manager(operation::copy, &other, this); } }
	any(any&& other) noexcept : storage { } { if (other.manager) { other.manager(operation::move, &other, this); } }

	template <typename T, typename Held = detail::stored_t<T>,
		typename = std::enable_if_t<!std::is_same<Held, any>::value && std::is_copy_constructible<Held>::value>>
	any(T&& value) : storage { }
	{
		// (a __restrict__ pointer converts to the plain pointer type it is stored as)
		if constexpr (fits<Held>) { ::new (storage.in_place) Held(static_cast<Held>(std::forward<T>(value))); }
		else { storage.on_heap = new Held(std::forward<T>(value)); }
		manager = &manage<Held>;
	}

	~any() { reset(); }

	any& operator=(const any& other) { any(other).swap(*this); return *this; }
	any& operator=(any&& other) noexcept { reset(); if (other.manager) { other.manager(operation::move, &other, this); } return *this; }
	template <typename T, typename = std::enable_if_t<!std::is_same<detail::stored_t<T>, any>::value>>
	any& operator=(T&& value) { any(std::forward<T>(value)).swap(*this); return *this; }

	void reset() noexcept { if (manager) { manager(operation::destroy, this, nullptr); manager = nullptr; } }
	void clear() noexcept { reset(); }                       // boost spelling
	bool has_value() const noexcept { return manager != nullptr; }
	bool empty() const noexcept { return manager == nullptr; } // boost spelling
	const std::type_info& type() const noexcept
	{
		return manager ? *static_cast<const std::type_info*>(manager(operation::type, this, nullptr)) : typeid(void);
	}
	void swap(any& other) noexcept
	{
		any tmp(std::move(other));
		other = std::move(*this);
		*this = std::move(tmp);
	}
};

/// pointer to the held value if `operand` holds exactly a (cv-, restrict-less) T, else nullptr
template <typename T>
inline const T* any_cast(const any* operand) noexcept
{
	using held = detail::stored_t<T>;
	if (!operand || !operand->manager || operand->manager != &any::manage<held>) {
		// the same type instantiated in another shared object has another manager address: compare type_info too
		if (!operand || !operand->manager || operand->type() != typeid(held)) { return nullptr; }
	}
	return static_cast<const T*>(operand->manager(any::operation::address, operand, nullptr));
}

template <typename T>
inline T* any_cast(any* operand) noexcept { return const_cast<T*>(any_cast<T>(static_cast<const any*>(operand))); }

template <typename T>
inline T any_cast(const any& operand)
{
	using value = std::remove_cv_t<std::remove_reference_t<T>>;
	auto held = any_cast<value>(&operand);
	if (!held) { throw bad_any_cast(); }
	return static_cast<T>(*held);
}

template <typename T>
inline optional<T> any_cast(const optional<any>& operand)
{
	return operand ? optional<T>(any_cast<T>(operand.value())) : optional<T>(nullopt);
}

template <typename Map>
inline optional<typename Map::mapped_type> maybe_at(const Map& map, const typename Map::key_type& key)
{
	auto it = map.find(key);
	return it == map.end() ? optional<typename Map::mapped_type>(nullopt) : optional<typename Map::mapped_type>(it->second);
}

} // namespace cuda
