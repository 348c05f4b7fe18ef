// String-keyed subclass factory + the FactoryProducible mix-in.
// Public surface follows the reference (src/util/Factory.h:36-101, src/util/FactoryProducible.h:25-85):
// registerClass / produce / canProduce / getInsantiators (sic), registerInSubclassFactory /
// getSubclassFactory / produceSubclass, with the same exception types and the same
// "No subclass of the base type ... is registered with key ..." message.
#pragma once
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <utility>

#include "giddy/util/type_name.hpp"

namespace util {

template <typename Key, typename T, typename... ConstructionArgs>
class Factory {
public:
	using Instantiator = T* (*)(ConstructionArgs...);
	using Instantiators = std::unordered_map<Key, Instantiator>;

	template <typename U>
	void registerClass(const Key& key, bool ignore_repeat_registration = false)
	{
		static_assert(std::is_base_of<T, U>::value, "U must derive from the factory's base class");
		if (subclassInstantiators.count(key) != 0) {
			if (ignore_repeat_registration) { return; }
			throw std::logic_error("Repeat registration of the same subclass in this factory.");
		}
		subclassInstantiators.emplace(key, +[](ConstructionArgs... args) -> T* { return new U(std::forward<ConstructionArgs>(args)...); });
	}

	std::unique_ptr<T> produce(const Key& subclass_key, ConstructionArgs... args) const
	{
		auto it = subclassInstantiators.find(subclass_key);
		if (it == subclassInstantiators.end()) {
			throw std::invalid_argument("Attempt to produce an instance of a class by a key not registered in the factory");
		}
		return std::unique_ptr<T>(it->second(std::forward<ConstructionArgs>(args)...));
	}

	bool canProduce(const Key& subclass_key) const { return subclassInstantiators.count(subclass_key) != 0; }

protected:
	Instantiators subclassInstantiators;
};

template <typename Key, typename T, typename... ConstructionArgs>
class ExposedFactory : public Factory<Key, T, ConstructionArgs...> {
public:
	using Instantiators = typename Factory<Key, T, ConstructionArgs...>::Instantiators;
	const Instantiators& getInsantiators() const { return this->subclassInstantiators; } // spelling as in Factory.h:98
	const Instantiators& getInstantiators() const { return this->subclassInstantiators; }
};

namespace mixins {

template <typename Key, typename Base, typename... ConstructionArgs>
class FactoryProducible {
public:
	using SubclassFactory = util::ExposedFactory<Key, Base, ConstructionArgs...>;

	template <typename U>
	static void registerInSubclassFactory(const Key& key, bool ignore_repeat_registration = true)
	{
		getSubclassFactory_().template registerClass<U>(key, ignore_repeat_registration);
	}

	static const SubclassFactory& getSubclassFactory() { return getSubclassFactory_(); }

	static std::unique_ptr<Base> produceSubclass(const Key& subclass_key, ConstructionArgs... args)
	{
		if (!getSubclassFactory().canProduce(subclass_key)) {
			throw std::invalid_argument(std::string("No subclass of the base type ")
				+ util::type_name<Base>() + " is registered with key \"" + std::string(subclass_key) + "\"");
		}
		return getSubclassFactory().produce(subclass_key, args...);
	}

	static bool canProduce(const Key& subclass_key) { return getSubclassFactory().canProduce(subclass_key); }

	FactoryProducible(const FactoryProducible&) = delete;
	FactoryProducible() = default;
	~FactoryProducible() = default;

protected:
	// function-local static: initialised on first use, so registration order across
	// translation units cannot hit an unconstructed factory
	static SubclassFactory& getSubclassFactory_()
	{
		static SubclassFactory subclass_factory;
		return subclass_factory;
	}
};

} // namespace mixins
} // namespace util
