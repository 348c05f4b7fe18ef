// C ABI of libgiddy_amd.so (declared in include/giddy.h): a non-throwing facade over the C++
// factory / wrapper API.  No compute happens here; every call forwards to one method of
// cuda::registered::kernel_t.
#include "giddy.h"

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "c_abi_internal.hpp"
#include "giddy/stream_scratch.hpp"

namespace giddy_internal {

std::string& last_error() { thread_local std::string message; return message; }

} // namespace giddy_internal

namespace {

using giddy_internal::guarded;

const std::vector<std::string>& sorted_keys()
{
	static const std::vector<std::string> keys = [] {
		std::vector<std::string> k;
		for (const auto& entry : cuda::registered::kernel_t::getSubclassFactory().getInsantiators()) { k.push_back(entry.first); }
		std::sort(k.begin(), k.end());
		return k;
	}();
	return keys;
}

template <typename T, size_t N>
cuda::array<T, N> coefficients_from(const T* values)
{
	cuda::array<T, N> a;
	for (size_t i = 0; i < N; i++) { a[i] = values[i]; }
	return a;
}

} // namespace

namespace giddy_internal {

cuda::any to_any(const giddy_argument_t& a)
{
	void* p = a.value.pointer;
	switch (a.type) {
	case GIDDY_BOOL: return cuda::any((bool) (a.value.u64 != 0));
	case GIDDY_U8:   return cuda::any((uint8_t)  a.value.u64);
	case GIDDY_U16:  return cuda::any((uint16_t) a.value.u64);
	case GIDDY_U32:  return cuda::any((uint32_t) a.value.u64);
	case GIDDY_U64:  return cuda::any((uint64_t) a.value.u64);
	case GIDDY_I8:   return cuda::any((int8_t)   a.value.i64);
	case GIDDY_I16:  return cuda::any((int16_t)  a.value.i64);
	case GIDDY_I32:  return cuda::any((int32_t)  a.value.i64);
	case GIDDY_I64:  return cuda::any((int64_t)  a.value.i64);
	case GIDDY_PTR_U8:   return cuda::any(static_cast<uint8_t*>(p));
	case GIDDY_PTR_U16:  return cuda::any(static_cast<uint16_t*>(p));
	case GIDDY_PTR_U32:  return cuda::any(static_cast<uint32_t*>(p));
	case GIDDY_PTR_U64:  return cuda::any(static_cast<uint64_t*>(p));
	case GIDDY_PTR_I8:   return cuda::any(static_cast<int8_t*>(p));
	case GIDDY_PTR_I16:  return cuda::any(static_cast<int16_t*>(p));
	case GIDDY_PTR_I32:  return cuda::any(static_cast<int32_t*>(p));
	case GIDDY_PTR_I64:  return cuda::any(static_cast<int64_t*>(p));
	case GIDDY_CPTR_U8:  return cuda::any(static_cast<const uint8_t*>(p));
	case GIDDY_CPTR_U16: return cuda::any(static_cast<const uint16_t*>(p));
	case GIDDY_CPTR_U32: return cuda::any(static_cast<const uint32_t*>(p));
	case GIDDY_CPTR_U64: return cuda::any(static_cast<const uint64_t*>(p));
	case GIDDY_CPTR_I8:  return cuda::any(static_cast<const int8_t*>(p));
	case GIDDY_CPTR_I16: return cuda::any(static_cast<const int16_t*>(p));
	case GIDDY_CPTR_I32: return cuda::any(static_cast<const int32_t*>(p));
	case GIDDY_CPTR_I64: return cuda::any(static_cast<const int64_t*>(p));
	case GIDDY_COEF_I16X1: return cuda::any(coefficients_from<int16_t, 1>(a.value.i16));
	case GIDDY_COEF_I16X2: return cuda::any(coefficients_from<int16_t, 2>(a.value.i16));
	case GIDDY_COEF_I32X1: return cuda::any(coefficients_from<int32_t, 1>(a.value.i32));
	case GIDDY_COEF_I32X2: return cuda::any(coefficients_from<int32_t, 2>(a.value.i32));
	case GIDDY_CPTR_COEF_I16X1: return cuda::any(static_cast<const cuda::array<int16_t, 1>*>(p));
	case GIDDY_CPTR_COEF_I16X2: return cuda::any(static_cast<const cuda::array<int16_t, 2>*>(p));
	case GIDDY_CPTR_COEF_I32X1: return cuda::any(static_cast<const cuda::array<int32_t, 1>*>(p));
	case GIDDY_CPTR_COEF_I32X2: return cuda::any(static_cast<const cuda::array<int32_t, 2>*>(p));
	default: throw std::invalid_argument("unknown giddy_type_t tag " + std::to_string(a.type) + " for argument \"" + (a.name ? a.name : "(null)") + "\"");
	}
}

cuda::kernel_t::arguments_type to_map(const giddy_argument_t* arguments, size_t n)
{
	if (arguments == nullptr && n != 0) { throw std::invalid_argument("null argument array"); }
	cuda::kernel_t::arguments_type map;
	for (size_t i = 0; i < n; i++) {
		if (arguments[i].name == nullptr) { throw std::invalid_argument("argument without a name"); }
		map[arguments[i].name] = to_any(arguments[i]);
	}
	return map;
}

} // namespace giddy_internal

namespace {

using giddy_internal::to_map;

const giddy_kernel& checked(const giddy_kernel_t* kernel)
{
	if (kernel == nullptr || !kernel->wrapper) { throw std::invalid_argument("null kernel handle"); }
	return *kernel;
}

} // namespace

extern "C" {

int giddy_kernel_count(void) { return (int) sorted_keys().size(); }

const char* giddy_kernel_key(int index)
{
	const auto& keys = sorted_keys();
	return (index < 0 || (size_t) index >= keys.size()) ? nullptr : keys[index].c_str();
}

int giddy_can_produce(const char* key)
{
	return key != nullptr && cuda::registered::kernel_t::getSubclassFactory().canProduce(key) ? 1 : 0;
}

int giddy_produce(const char* key, giddy_kernel_t** kernel)
{
	return guarded([&] {
		if (key == nullptr || kernel == nullptr) { throw std::invalid_argument("null key or output handle"); }
		auto wrapper = cuda::registered::kernel_t::produceSubclass(key);
		*kernel = new giddy_kernel { std::shared_ptr<const cuda::registered::kernel_t>(std::move(wrapper)) };
	});
}

void giddy_release(giddy_kernel_t* kernel) { delete kernel; }

int giddy_resolve_launch_configuration(
	const giddy_kernel_t* kernel, int device_id,
	const giddy_argument_t* arguments, size_t num_arguments,
	const giddy_launch_configuration_limits_t* limits,
	giddy_launch_configuration_t* launch_configuration)
{
	return guarded([&] {
		const auto& k = checked(kernel);
		if (launch_configuration == nullptr) { throw std::invalid_argument("null output launch configuration"); }
		cuda::launch_configuration_limits_t l { cuda::nullopt, cuda::nullopt, cuda::nullopt };
		if (limits != nullptr) {
			if (limits->has_block_size) { l.block_size = limits->block_size; }
			if (limits->has_grid_size) { l.grid_size = limits->grid_size; }
			if (limits->has_dynamic_shared_memory) { l.dynamic_shared_memory = limits->dynamic_shared_memory; }
		}
		auto map = to_map(arguments, num_arguments);
		cuda::launch_configuration_t config;
		if (device_id == GIDDY_NOMINAL_MI355X_DEVICE) {
			config = k.wrapper->resolve_launch_configuration(
				cuda::device::properties_t::nominal_mi355x(), cuda::device_function::attributes_t(), map, l);
		}
		else {
			if (device_id == GIDDY_CURRENT_DEVICE) { device_id = cuda::device::current::get().id(); }
			auto properties = cuda::device::get(device_id).properties();
			cuda::device::current::scoped_override_t<> use(device_id);
			config = k.wrapper->resolve_launch_configuration(properties, k.wrapper->get_device_function().attributes(), map, l);
		}
		launch_configuration->grid_dimensions[0] = config.grid_dimensions.x;
		launch_configuration->grid_dimensions[1] = config.grid_dimensions.y;
		launch_configuration->grid_dimensions[2] = config.grid_dimensions.z;
		launch_configuration->block_dimensions[0] = config.block_dimensions.x;
		launch_configuration->block_dimensions[1] = config.block_dimensions.y;
		launch_configuration->block_dimensions[2] = config.block_dimensions.z;
		launch_configuration->dynamic_shared_memory_size = config.dynamic_shared_memory_size;
	});
}

int giddy_enqueue_launch(
	const giddy_kernel_t* kernel, void* hip_stream,
	const giddy_launch_configuration_t* launch_configuration,
	const giddy_argument_t* arguments, size_t num_arguments)
{
	return guarded([&] {
		const auto& k = checked(kernel);
		if (launch_configuration == nullptr) { throw std::invalid_argument("null launch configuration"); }
		cuda::launch_configuration_t config(
			cuda::dimensions_t(launch_configuration->grid_dimensions[0], launch_configuration->grid_dimensions[1], launch_configuration->grid_dimensions[2]),
			cuda::dimensions_t(launch_configuration->block_dimensions[0], launch_configuration->block_dimensions[1], launch_configuration->block_dimensions[2]),
			launch_configuration->dynamic_shared_memory_size);
		k.wrapper->enqueue_launch(static_cast<hipStream_t>(hip_stream), config, to_map(arguments, num_arguments));
	});
}

const void* giddy_device_function(const giddy_kernel_t* kernel)
{
	return (kernel != nullptr && kernel->wrapper) ? kernel->wrapper->get_device_function().ptr() : nullptr;
}

int giddy_device_function_count(const giddy_kernel_t* kernel)
{
	return (kernel != nullptr && kernel->wrapper) ? 1 + (int) kernel->wrapper->get_device_function().num_alternates() : 0;
}

const void* giddy_device_function_alternate(const giddy_kernel_t* kernel, int index)
{
	if (kernel == nullptr || !kernel->wrapper || index < 0) { return nullptr; }
	const auto function = kernel->wrapper->get_device_function();
	return index == 0 ? function.ptr() : function.alternate((unsigned) index - 1);
}

int giddy_prepare(const giddy_kernel_t* kernel, const giddy_argument_t* arguments, size_t num_arguments, giddy_prepared_t** prepared)
{
	return guarded([&] {
		const auto& k = checked(kernel);
		if (prepared == nullptr) { throw std::invalid_argument("null output handle"); }
		*prepared = new giddy_prepared { k.wrapper, to_map(arguments, num_arguments) };
	});
}

int giddy_prepared_update(giddy_prepared_t* prepared, const giddy_argument_t* arguments, size_t num_arguments)
{
	return guarded([&] {
		if (prepared == nullptr || prepared->wrapper == nullptr) { throw std::invalid_argument("null prepared launch"); }
		if (arguments == nullptr && num_arguments != 0) { throw std::invalid_argument("null argument array"); }
		for (size_t i = 0; i < num_arguments; i++) {
			if (arguments[i].name == nullptr) { throw std::invalid_argument("argument without a name"); }
			prepared->arguments[arguments[i].name] = giddy_internal::to_any(arguments[i]);
		}
	});
}

int giddy_enqueue_prepared(const giddy_prepared_t* prepared, void* hip_stream, const giddy_launch_configuration_t* launch_configuration)
{
	return guarded([&] {
		if (prepared == nullptr || prepared->wrapper == nullptr) { throw std::invalid_argument("null prepared launch"); }
		if (launch_configuration == nullptr) { throw std::invalid_argument("null launch configuration"); }
		const cuda::launch_configuration_t config(
			cuda::dimensions_t(launch_configuration->grid_dimensions[0], launch_configuration->grid_dimensions[1], launch_configuration->grid_dimensions[2]),
			cuda::dimensions_t(launch_configuration->block_dimensions[0], launch_configuration->block_dimensions[1], launch_configuration->block_dimensions[2]),
			launch_configuration->dynamic_shared_memory_size);
		prepared->wrapper->enqueue_launch_with(static_cast<hipStream_t>(hip_stream), config, prepared->arguments);
	});
}

void giddy_release_prepared(giddy_prepared_t* prepared) { delete prepared; }

// ---- batches (include/giddy/batch.hpp) --------------------------------------------------------------------------------

int giddy_can_launch_batches(const giddy_kernel_t* kernel)
{
	return (kernel != nullptr && kernel->wrapper && kernel->wrapper->launches_batches()) ? 1 : 0;
}

int giddy_batch_create(const giddy_kernel_t* kernel, int device_id, giddy_batch_t** batch)
{
	return guarded([&] {
		const auto& k = checked(kernel);
		if (batch == nullptr) { throw std::invalid_argument("null output handle"); }
		if (!k.wrapper->launches_batches()) { throw std::logic_error("this kernel wrapper does not launch batches"); }
		std::unique_ptr<giddy_batch> b(new giddy_batch);
		b->wrapper = k.wrapper;
		if (device_id == GIDDY_NOMINAL_MI355X_DEVICE) {
			b->properties = cuda::device::properties_t::nominal_mi355x();
			b->attributes = cuda::device_function::attributes_t();
		}
		else {
			if (device_id == GIDDY_CURRENT_DEVICE) { device_id = cuda::device::current::get().id(); }
			b->properties = cuda::device::get(device_id).properties();
			cuda::device::current::scoped_override_t<> use(device_id);
			b->attributes = k.wrapper->get_device_function().attributes();
		}
		b->device_id = device_id;
		*batch = b.release();
	});
}

int giddy_batch_add(giddy_batch_t* batch, const giddy_argument_t* launch_arguments, size_t num_launch_arguments,
	const giddy_argument_t* resolution_arguments, size_t num_resolution_arguments)
{
	return guarded([&] {
		if (batch == nullptr || !batch->wrapper) { throw std::invalid_argument("null batch"); }
		const auto piece_config = batch->wrapper->resolve_launch_configuration(batch->properties, batch->attributes,
			to_map(resolution_arguments, num_resolution_arguments), cuda::launch_configuration_limits_t { cuda::nullopt, cuda::nullopt, cuda::nullopt });
		batch->wrapper->add_to_batch(batch->table, piece_config, to_map(launch_arguments, num_launch_arguments));
		batch->uploaded = false;
	});
}

int giddy_batch_clear(giddy_batch_t* batch)
{
	return guarded([&] {
		if (batch == nullptr) { throw std::invalid_argument("null batch"); }
		batch->table.clear();
		batch->uploaded = false;
	});
}

uint64_t giddy_batch_num_pieces(const giddy_batch_t* batch) { return batch != nullptr ? (uint64_t) batch->table.pieces().size() : 0; }
uint64_t giddy_batch_num_blocks(const giddy_batch_t* batch) { return batch != nullptr ? (uint64_t) batch->table.blocks().size() : 0; }

int giddy_batch_upload(giddy_batch_t* batch, void* hip_stream)
{
	return guarded([&] {
		if (batch == nullptr || !batch->wrapper) { throw std::invalid_argument("null batch"); }
		if (batch->device_id == GIDDY_NOMINAL_MI355X_DEVICE) { throw std::logic_error("a batch created for the nominal device has no device to be uploaded to"); }
		// (the tables go to the device the batch was created for — `hip_stream` must be one of its streams —, whichever device is current for the caller)
		cuda::device::current::scoped_override_t<> on_the_batchs_device(batch->device_id);
		batch->on_device.upload(batch->table, static_cast<hipStream_t>(hip_stream));
		batch->uploaded = true;
	});
}

int giddy_enqueue_batch(giddy_batch_t* batch, void* hip_stream)
{
	return guarded([&] {
		if (batch == nullptr || !batch->wrapper) { throw std::invalid_argument("null batch"); }
		if (batch->device_id == GIDDY_NOMINAL_MI355X_DEVICE) { throw std::logic_error("a batch created for the nominal device cannot be launched"); }
		cuda::device::current::scoped_override_t<> on_the_batchs_device(batch->device_id);
		if (!batch->uploaded) {
			// (allocates device memory: inside a stream capture call giddy_batch_upload BEFORE the capture begins — include/giddy.h)
			batch->on_device.upload(batch->table, static_cast<hipStream_t>(hip_stream));
			batch->uploaded = true;
		}
		if (batch->on_device.view().num_blocks == 0) { return; }
		batch->wrapper->enqueue_launch_of_batch(static_cast<hipStream_t>(hip_stream), batch->on_device.view());
	});
}

int giddy_batch_tables(const giddy_batch_t* batch, giddy_batch_tables_t* tables)
{
	return guarded([&] {
		if (batch == nullptr || tables == nullptr) { throw std::invalid_argument("null batch or output"); }
		if (!batch->uploaded) { throw std::logic_error("the batch has not been uploaded since it was last changed"); }
		const auto& view = batch->on_device.view();
		tables->pieces = reinterpret_cast<const giddy_batch_piece_t*>(view.pieces);
		tables->blocks = reinterpret_cast<const giddy_batch_block_t*>(view.blocks);
		tables->num_pieces = view.num_pieces; tables->num_blocks = view.num_blocks;
		tables->block_size = view.block_size; tables->dynamic_shared_memory_size = view.dynamic_shared_memory_size;
		for (unsigned i = 0; i < cuda::batch::hints_per_batch; i++) { tables->hints[i] = view.hints[i]; }
	});
}

int giddy_enqueue_batch_of_device_tables(const giddy_kernel_t* kernel, void* hip_stream, const giddy_batch_tables_t* tables)
{
	return guarded([&] {
		const auto& k = checked(kernel);
		if (tables == nullptr) { throw std::invalid_argument("null tables"); }
		if (tables->num_blocks == 0) { return; }
		if (tables->pieces == nullptr || tables->blocks == nullptr) { throw std::invalid_argument("null piece or block table"); }
		if (tables->block_size == 0 || tables->block_size > 1024) { throw std::invalid_argument("a batch's block size must be 1 ... 1024 threads"); }
		cuda::batch::on_device_t view;
		view.pieces = reinterpret_cast<const cuda::batch::piece_t*>(tables->pieces);
		view.blocks = reinterpret_cast<const cuda::batch::block_t*>(tables->blocks);
		view.num_pieces = tables->num_pieces; view.num_blocks = tables->num_blocks;
		view.block_size = tables->block_size; view.dynamic_shared_memory_size = tables->dynamic_shared_memory_size;
		for (unsigned i = 0; i < cuda::batch::hints_per_batch; i++) { view.hints[i] = tables->hints[i]; }
		k.wrapper->enqueue_launch_of_batch(static_cast<hipStream_t>(hip_stream), view);
	});
}

void giddy_batch_release(giddy_batch_t* batch) { delete batch; }

int giddy_reserve_scratch(void* hip_stream, uint64_t bytes)
{
	return guarded([&] {
		if (!cuda::stream_scratch::reserve(static_cast<hipStream_t>(hip_stream), (size_t) bytes)) { throw cuda::runtime_error(hipErrorOutOfMemory, "no device memory for the scratch asked for"); }
	});
}

int giddy_release_scratch(void* hip_stream)
{
	return guarded([&] { cuda::stream_scratch::forget(static_cast<hipStream_t>(hip_stream)); });
}

uint64_t giddy_trim(void) { return (uint64_t) cuda::stream_scratch::trim(); }

uint64_t giddy_scratch_held_bytes(void) { return (uint64_t) cuda::stream_scratch::held_bytes(); }

int giddy_set_capture_scratch_policy(int policy)
{
	return guarded([&] {
		if (policy != 0 && policy != 1) { throw std::invalid_argument("capture scratch policy: 0 (none) or 1 (graph allocation)"); }
		cuda::stream_scratch::set_capture_policy(policy == 1 ? cuda::stream_scratch::capture_policy::graph_allocation : cuda::stream_scratch::capture_policy::none);
	});
}

const char* giddy_last_error(void) { return giddy_internal::last_error().c_str(); }

const char* giddy_version(void) { return "libgiddy_amd 0.5 (gfx950)"; }

} // extern "C"
