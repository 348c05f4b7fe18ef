/*
 * giddy.h — C ABI of libgiddy_amd.so: the string-keyed kernel-wrapper factory, launch-configuration
 * resolution and asynchronous enqueue, for callers that cannot instantiate the C++ templates
 * (ctypes / cgo / JNI style bindings, and this repository's own parity tests and bench).
 *
 * The reference (eyalroz/libgiddy) has NO C ABI or FFI layer: its boundary is the C++ API in
 * include/giddy/ (same names as the reference's src/kernels and src/kernel_wrappers).  Each entry
 * point below is a thin, non-throwing facade over one call of that API, so that what a binding
 * does reads like the reference's "mode of use A" — factory-provided, type-erased wrappers
 * (examples/src/modes_of_use/factory_provided_type_erased_wrapper.cu:7-45):
 *
 *   giddy_produce                       <-  cuda::registered::kernel_t::produceSubclass(key)
 *                                           src/kernel_wrappers/registered_wrapper.h:28-39,
 *                                           src/util/FactoryProducible.h:63-70
 *   giddy_can_produce                   <-  getSubclassFactory().canProduce(key)   src/util/Factory.h:81-83
 *   giddy_kernel_count / giddy_kernel_key
 *                                       <-  getSubclassFactory().getInsantiators() src/util/Factory.h:97-100
 *   giddy_resolve_launch_configuration  <-  kernel_t::resolve_launch_configuration(device_properties,
 *                                           kernel_function_attributes, arguments_type, limits)
 *                                           src/cuda/kernel_wrapper.cuh:89-121
 *   giddy_enqueue_launch                <-  kernel_t::enqueue_launch(stream, launch_config, arguments_type)
 *                                           src/cuda/kernel_wrapper.cuh:67-70
 *   giddy_device_function               <-  kernel_t::get_device_function()        src/cuda/kernel_wrapper.cuh:123
 *
 * Type-erased arguments: the C++ wrappers any_cast every argument to an EXACT C++ type
 * (src/cuda/kernel_wrapper.cuh:47-49, :228-235); giddy_argument_t carries that type as a tag.
 * Argument names and types per scheme are tabulated in INTEGRATION.md.
 *
 * Conventions: every function returns GIDDY_SUCCESS (0) or a negative giddy_status_t and never
 * throws; giddy_last_error() returns the failed call's message (thread-local).  Pointers passed in
 * arguments are DEVICE pointers owned by the caller; launches are asynchronous on the given HIP
 * stream of the current device and the library never synchronises.
 */
#ifndef GIDDY_H_
#define GIDDY_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct giddy_kernel giddy_kernel_t; /* opaque: one cuda::registered::kernel_t instance */

typedef enum giddy_status {
	GIDDY_SUCCESS            =  0,
	GIDDY_INVALID_ARGUMENT   = -1, /* std::invalid_argument (unknown key, unsatisfiable launch constraints, bad segment length ...) */
	GIDDY_LOGIC_ERROR        = -2, /* std::logic_error (e.g. a serialization factor the kernel cannot accept) */
	GIDDY_OUT_OF_RANGE       = -3, /* std::out_of_range: a required argument name is missing from the map */
	GIDDY_BAD_ANY_CAST       = -4, /* an argument holds a different C++ type than the wrapper expects */
	GIDDY_RUNTIME_ERROR      = -5, /* the HIP runtime reported an error */
	GIDDY_UNKNOWN_ERROR      = -6
} giddy_status_t;

/* The C++ type an argument holds.  size_t is uint64_t, unsigned is uint32_t, unsigned char is
 * uint8_t and serialization_factor_t is uint16_t on this platform, so those share tags. */
typedef enum giddy_type {
	GIDDY_BOOL = 1,
	GIDDY_U8, GIDDY_U16, GIDDY_U32, GIDDY_U64,            /* by-value unsigned scalars (GIDDY_U64 == size_t) */
	GIDDY_I8, GIDDY_I16, GIDDY_I32, GIDDY_I64,            /* by-value signed scalars */
	GIDDY_PTR_U8, GIDDY_PTR_U16, GIDDY_PTR_U32, GIDDY_PTR_U64,       /* T*        (outputs) */
	GIDDY_PTR_I8, GIDDY_PTR_I16, GIDDY_PTR_I32, GIDDY_PTR_I64,
	GIDDY_CPTR_U8, GIDDY_CPTR_U16, GIDDY_CPTR_U32, GIDDY_CPTR_U64,   /* const T*  (inputs) */
	GIDDY_CPTR_I8, GIDDY_CPTR_I16, GIDDY_CPTR_I32, GIDDY_CPTR_I64,
	/* model coefficient records, cuda::array<T, N>: by value (Model) and const pointer to records (FOR) */
	GIDDY_COEF_I16X1, GIDDY_COEF_I16X2, GIDDY_COEF_I32X1, GIDDY_COEF_I32X2,
	GIDDY_CPTR_COEF_I16X1, GIDDY_CPTR_COEF_I16X2, GIDDY_CPTR_COEF_I32X1, GIDDY_CPTR_COEF_I32X2
} giddy_type_t;

typedef struct giddy_argument {
	const char*  name;   /* the wrapper's argument-map key, e.g. "compressed_input" */
	int32_t      type;   /* giddy_type_t */
	int32_t      reserved;
	union {
		uint64_t  u64;       /* scalars, zero- or sign-extended */
		int64_t   i64;
		void*     pointer;   /* device pointer */
		int32_t   i32[2];    /* GIDDY_COEF_I32X{1,2} */
		int16_t   i16[2];    /* GIDDY_COEF_I16X{1,2} */
	} value;
} giddy_argument_t;

/* cuda::launch_configuration_t */
typedef struct giddy_launch_configuration {
	uint32_t  grid_dimensions[3];
	uint32_t  block_dimensions[3];
	uint32_t  dynamic_shared_memory_size;
} giddy_launch_configuration_t;

/* cuda::launch_configuration_limits_t (src/cuda/kernel_wrapper.cuh:20-33); has_* = optional engaged */
typedef struct giddy_launch_configuration_limits {
	int32_t   has_block_size;             uint32_t  block_size;
	int32_t   has_grid_size;              uint32_t  grid_size;
	int32_t   has_dynamic_shared_memory;  uint32_t  dynamic_shared_memory;
} giddy_launch_configuration_limits_t;

#define GIDDY_NOMINAL_MI355X_DEVICE (-1)
#define GIDDY_CURRENT_DEVICE        (-2)

/* ---- factory ---------------------------------------------------------------------------- */
int          giddy_kernel_count(void);                 /* number of registered wrapper instantiations (the reference's 163 + 57 fused decompress-and-patch extensions + 8 expansions of compressed patch positions + 110 device-side compressors) */
const char*  giddy_kernel_key(int index);              /* keys in lexicographic order; NULL if out of range */
int          giddy_can_produce(const char* key);       /* 1 / 0 */
int          giddy_produce(const char* key, giddy_kernel_t** kernel);
void         giddy_release(giddy_kernel_t* kernel);

/* ---- per-kernel --------------------------------------------------------------------------- */
/* device_id >= 0: properties and function attributes are queried from that HIP device;
 * device_id == GIDDY_CURRENT_DEVICE: from the calling thread's current HIP device.
 * device_id == GIDDY_NOMINAL_MI355X_DEVICE: nominal MI355X properties, no device is touched
 * (host-only resolution, usable where no GPU is present). */
int          giddy_resolve_launch_configuration(
                 const giddy_kernel_t* kernel, int device_id,
                 const giddy_argument_t* arguments, size_t num_arguments,
                 const giddy_launch_configuration_limits_t* limits /* may be NULL */,
                 giddy_launch_configuration_t* launch_configuration);

int          giddy_enqueue_launch(
                 const giddy_kernel_t* kernel, void* hip_stream,
                 const giddy_launch_configuration_t* launch_configuration,
                 const giddy_argument_t* arguments, size_t num_arguments);

const void*  giddy_device_function(const giddy_kernel_t* kernel);

/* ---- diagnostics -------------------------------------------------------------------------- */
const char*  giddy_last_error(void);                   /* message of this thread's last failed call */
const char*  giddy_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GIDDY_H_ */
