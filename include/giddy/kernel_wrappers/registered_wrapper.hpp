// cuda::registered::kernel_t — a kernel wrapper that registers itself in a string-keyed factory
// during static initialisation — and the per-wrapper boilerplate macro.  Interface and key
// construction follow the reference's src/kernel_wrappers/registered_wrapper.h:28-39,63-88:
// key = demangled type name with everything up to and including "cuda::kernels::" removed.
#pragma once
#include <string>

#include "giddy/kernel_wrapper.hpp"
#include "giddy/util/factory.hpp"
#include "giddy/util/type_name.hpp"

namespace cuda {
namespace registered {

class kernel_t : public cuda::kernel_t, protected util::mixins::FactoryProducible<std::string, kernel_t> {
public:
	using factory_key_type = std::string;
	using parent = cuda::kernel_t;
	using factory_producible_mixin_type = util::mixins::FactoryProducible<factory_key_type, kernel_t>;
	using factory_producible_mixin_type::produceSubclass;
	using factory_producible_mixin_type::getSubclassFactory;
	using factory_producible_mixin_type::canProduce;
};

namespace detail {
inline std::string factory_key_of(const std::string& type_name)
{
	static const std::string prefix = "cuda::kernels::";
	auto pos = type_name.find(prefix);
	return pos == std::string::npos ? type_name : type_name.substr(pos + prefix.length());
}
} // namespace detail

} // namespace registered
} // namespace cuda

// What every wrapper class declares: its own factory registration (keyed by its demangled type name)
// and the three virtual overrides.  The member signatures are the API of cuda::kernel_t.
#define REGISTERED_KERNEL_WRAPPER_BOILERPLATE_DEFINITIONS(subclass_name) \
	using parent = cuda::registered::kernel_t; \
	using arguments_type = parent::arguments_type; \
	using parent::resolve_launch_configuration; \
	static void registerInSubclassFactory() \
	{ \
		const std::string key = cuda::registered::detail::factory_key_of(util::type_name<subclass_name>()); \
		FactoryProducible::registerInSubclassFactory<subclass_name>(key); \
	} \
public: \
	launch_configuration_t resolve_launch_configuration(device::properties_t device_properties, \
		device_function::attributes_t kernel_function_attributes, arguments_type extra_arguments, \
		launch_configuration_limits_t limits = { nullopt, nullopt, nullopt }) const override final; \
	/* the stream is assumed to belong to the current device */ \
	void enqueue_launch(stream::id_t stream, const launch_configuration_t& launch_config, \
		arguments_type arguments) const override final { enqueue_launch_with(stream, launch_config, arguments); } \
	/* ... the same, reading the caller's map where it is (what each wrapper defines) */ \
	void enqueue_launch_with(stream::id_t stream, const launch_configuration_t& launch_config, \
		const arguments_type& arguments) const override final; \
	const device_function_t get_device_function() const override final

// For a wrapper with ONE typed entry point `static void launch(stream, launch_config, typed parameters...)` — the function its type-erased
// enqueue_launch() calls after unpacking the map, and the one cuda::kernel::enqueue_launch(kernel, ...) finds for a caller who
// instantiated the wrapper (kernel_wrapper.hpp): a caller who holds only the base type reaches it too, through this override.
#define REGISTERED_KERNEL_WRAPPER_TYPED_LAUNCH(subclass_name) \
public: \
	void enqueue_launch_of_packed_parameters(stream::id_t stream_id, const launch_configuration_t& launch_config, \
		void* const* parameters, const size_t* sizes, size_t count) const override final \
	{ \
		cuda::kernel::detail::launch_packed(&subclass_name::launch, stream_id, launch_config, parameters, sizes, count); \
	}

// For a wrapper that decodes BATCHES (giddy/batch.hpp).  The wrapper defines
//   template <typename Sink> static void with_typed_parameters(const launch_configuration_t&, const arguments_type&, Sink&&)   the map unpacked: sink(typed parameters of launch()...)
//   static void add_piece(batch::table_t&, const launch_configuration_t&, typed parameters...)                              (this macro's default: table.add)
//   static void launch_batch(stream::id_t, const batch::on_device_t&)                                                        the ONE launch
// and this macro routes the three virtuals of cuda::kernel_t through them — enqueue_launch_with() included, so that the single launch and
// the piece of a batch read the caller's map by the same lines.
#define REGISTERED_KERNEL_WRAPPER_BATCHES(subclass_name) \
public: \
	bool launches_batches() const override final { return true; } \
	void add_to_batch(cuda::batch::table_t& table, const launch_configuration_t& piece_config, const arguments_type& piece_arguments) const override final \
	{ \
		with_typed_parameters(piece_config, piece_arguments, [&](auto... parameters) { subclass_name::add_piece(table, piece_config, parameters...); }); \
	} \
	void enqueue_launch_of_batch(stream::id_t stream, const cuda::batch::on_device_t& tables) const override final { subclass_name::launch_batch(stream, tables); }
