// What the translation units behind include/giddy.h share: the exception-to-status guard, the thread's last error message and the
// conversion of tagged C arguments into the wrappers' type-erased argument map.
#pragma once
#include "giddy.h"

#include <any>
#include <memory>
#include <stdexcept>
#include <string>

#include "giddy/array.hpp"
#include "giddy/kernel_wrappers/registered_wrapper.hpp"

struct giddy_kernel {
	std::shared_ptr<const cuda::registered::kernel_t> wrapper;
};

/// a wrapper with an argument map that is kept between launches (the wrapper is shared with the giddy_kernel_t it was prepared from: releasing that one first is fine)
struct giddy_prepared {
	std::shared_ptr<const cuda::registered::kernel_t> wrapper;
	cuda::kernel_t::arguments_type arguments;
};

/// a batch (include/giddy/batch.hpp): the tables as the host builds them, their copy on the device, the wrapper that launches them
struct giddy_batch {
	std::shared_ptr<const cuda::registered::kernel_t> wrapper;
	int device_id;                         // a HIP device, or GIDDY_NOMINAL_MI355X_DEVICE (pieces can be added and counted, not uploaded)
	cuda::device::properties_t properties;
	cuda::device_function::attributes_t attributes;
	cuda::batch::table_t table;
	cuda::batch::uploaded_t on_device;
	bool uploaded = false;                 // `on_device` holds `table` as it is now
};

namespace giddy_internal {

std::string& last_error();   // this thread's
cuda::any to_any(const giddy_argument_t& argument);
cuda::kernel_t::arguments_type to_map(const giddy_argument_t* arguments, size_t count);

template <typename F>
int guarded(F&& f) noexcept
{
	std::string& message = last_error();
	try { f(); message.clear(); return GIDDY_SUCCESS; }
	catch (const std::bad_any_cast& e)     { message = std::string("bad_any_cast: an argument does not hold the exact type the wrapper expects (") + e.what() + ")"; return GIDDY_BAD_ANY_CAST; }
	catch (const std::out_of_range& e)     { message = std::string("out_of_range (missing argument?): ") + e.what(); return GIDDY_OUT_OF_RANGE; }
	catch (const std::invalid_argument& e) { message = std::string("invalid_argument: ") + e.what(); return GIDDY_INVALID_ARGUMENT; }
	catch (const std::logic_error& e)      { message = std::string("logic_error: ") + e.what(); return GIDDY_LOGIC_ERROR; }
	catch (const cuda::runtime_error& e)   { message = std::string("HIP runtime error: ") + e.what(); return GIDDY_RUNTIME_ERROR; }
	catch (const std::exception& e)        { message = std::string("exception: ") + e.what(); return GIDDY_UNKNOWN_ERROR; }
	catch (...)                            { message = "unknown exception"; return GIDDY_UNKNOWN_ERROR; }
}

} // namespace giddy_internal
