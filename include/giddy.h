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
 * stream of the current device and the library never synchronises: no entry point that enqueues work waits for the
 * device (the first long column on a stream may ALLOCATE scratch — hipMalloc, which does not wait either; giddy_reserve_scratch
 * moves that out of the launch path).  giddy_trim() is the one call that waits, and says so.
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

/* The kernels the wrapper may launch WITH THE CALLER'S CONFIGURATION: index 0 is giddy_device_function() — the kernel of the reference's
 * signature —, the others the builds its typed launch picks among (run decoders: columns of short / long runs).  The attributes
 * giddy_resolve_launch_configuration resolves against are limits that hold for all of them (cuda::device_function_t::attributes()). */
int          giddy_device_function_count(const giddy_kernel_t* kernel);
const void*  giddy_device_function_alternate(const giddy_kernel_t* kernel, int index);

/* ---- prepared launches ---------------------------------------------------------------------
 * A loop that launches the same wrapper many times (a column's chunks): the argument map is built ONCE, entries are rewritten in place,
 * and a launch copies nothing.  Semantics of each launch are giddy_enqueue_launch's with the handle's current arguments
 * (cuda::kernel_t::enqueue_launch_with, include/giddy/kernel_wrapper.hpp — this library's addition to the reference's interface). */
typedef struct giddy_prepared giddy_prepared_t;
int          giddy_prepare(const giddy_kernel_t* kernel, const giddy_argument_t* arguments, size_t num_arguments, giddy_prepared_t** prepared);
int          giddy_prepared_update(giddy_prepared_t* prepared, const giddy_argument_t* arguments, size_t num_arguments); /* replaces (or adds) the entries named */
int          giddy_enqueue_prepared(const giddy_prepared_t* prepared, void* hip_stream, const giddy_launch_configuration_t* launch_configuration);
void         giddy_release_prepared(giddy_prepared_t* prepared);

/* (a giddy_prepared_t shares its wrapper with the giddy_kernel_t it was prepared from: either may be released first) */

/* ---- batches: MANY small columns, ONE launch ------------------------------------------------------
 * The reference launches once per column (src/cuda/kernel_wrapper.cuh:135-147) and exists to expand chunks of columns where they
 * arrive (README.md:15-24); a launch per small chunk costs the host microseconds and the device a ramp and a tail
 * (profiles/r05_size_sweep_round_start.txt: a column of 2^20 elements decodes at 5-20 % of the rate of one of 2^28).  A batch
 * (include/giddy/batch.hpp) is a table of PIECES — each a column as giddy_enqueue_launch would be given it — decoded by one launch of
 * the wrapper's batch kernel: pieces are independent, of any lengths (0 and 1 included), each with the workgroups its own launch
 * configuration would have had.  Wrappers that launch batches: Dictionary, Frame of Reference, Delta, both Discard-Zero-Bytes
 * forms, Run-Length and Run-Position encoding (giddy_can_launch_batches).
 *
 *   giddy_batch_create(kernel, device, &batch);
 *   for each piece:  giddy_batch_add(batch, launch arguments, n, resolution arguments, m);   <- host only: nothing is enqueued
 *   giddy_enqueue_batch(batch, stream);     <- one asynchronous copy of the tables (the first time, or after a change) + ONE launch
 *   ... giddy_enqueue_batch(batch, stream) again decodes the same pieces again; giddy_batch_clear() starts another table.
 *   giddy_batch_release(batch);             <- once the launches that read its tables have run (it frees device memory: waits)
 *
 * The tables are uploaded to, and the launch runs on, the device the batch was CREATED for, whichever device is current in the calling
 * thread: `hip_stream` must be a stream of that device.  The first upload ALLOCATES device memory — when the launch is to be captured
 * into a HIP graph, call giddy_batch_upload before the capture begins.
 *
 * Pieces want what single columns want: outputs on 128-byte lines and inputs aligned for wide loads (a piece elsewhere switches the
 * whole batch to the any-alignment build of Dictionary / Frame of Reference / fixed-width Discard-Zero-Bytes); the run decoders
 * pick their build by the mean run of the whole batch. */
typedef struct giddy_batch giddy_batch_t;

/* The tables where the device reads them — the layouts are a contract (cuda::batch::piece_t / block_t): a caller may also WRITE them on
 * the device (a kernel that cuts a column it has just received) and launch with giddy_enqueue_batch_of_device_tables. */
typedef struct giddy_batch_piece {
	uint64_t  slot[12];      /* parameter i of the kernel of the reference's signature, in its order, zero-extended (pointers: the address; bool: 0 / 1) */
	uint32_t  blocks;        /* workgroups of the launch that decode this piece (0: none) */
	uint32_t  first_block;   /* the first of them */
	uint64_t  reserved;
} giddy_batch_piece_t;       /* 112 bytes, 16-byte aligned */
typedef struct giddy_batch_block { uint32_t piece, block_in_piece; } giddy_batch_block_t;   /* workgroup w of the launch: entry w */
typedef struct giddy_batch_tables {
	const giddy_batch_piece_t*  pieces;                       /* device memory */
	const giddy_batch_block_t*  blocks;                       /* device memory, 8-byte aligned; num_blocks entries */
	uint32_t                    num_pieces, num_blocks;
	uint32_t                    block_size;                   /* threads per workgroup of the launch */
	uint32_t                    dynamic_shared_memory_size;
	uint64_t                    hints[4];                     /* run decoders: [0] outputs and [1] runs of all pieces, [2] != 0: some run array is not 16-byte aligned; [3] != 0: a piece needs the any-alignment build */
} giddy_batch_tables_t;

int          giddy_can_launch_batches(const giddy_kernel_t* kernel);   /* 1 / 0 */
int          giddy_batch_create(const giddy_kernel_t* kernel, int device_id /* as giddy_resolve_launch_configuration */, giddy_batch_t** batch);
/* one more piece: the launch arguments of giddy_enqueue_launch and the resolution arguments of giddy_resolve_launch_configuration for THAT column */
int          giddy_batch_add(giddy_batch_t* batch, const giddy_argument_t* launch_arguments, size_t num_launch_arguments,
                             const giddy_argument_t* resolution_arguments, size_t num_resolution_arguments);
int          giddy_batch_clear(giddy_batch_t* batch);
uint64_t     giddy_batch_num_pieces(const giddy_batch_t* batch);
uint64_t     giddy_batch_num_blocks(const giddy_batch_t* batch);       /* workgroups of the one launch */
int          giddy_batch_upload(giddy_batch_t* batch, void* hip_stream);   /* the tables to the device now (asynchronous, pinned staging); giddy_enqueue_batch does it when needed */
int          giddy_enqueue_batch(giddy_batch_t* batch, void* hip_stream);
int          giddy_batch_tables(const giddy_batch_t* batch, giddy_batch_tables_t* tables);   /* where an uploaded batch's tables lie */
int          giddy_enqueue_batch_of_device_tables(const giddy_kernel_t* kernel, void* hip_stream, const giddy_batch_tables_t* tables);
void         giddy_batch_release(giddy_batch_t* batch);

/* ---- scratch memory --------------------------------------------------------------------------
 * Wrappers whose fast paths need memory of their own between their launches (chained tiles of long Delta segments, finer anchors of long
 * run-encoded / discard-zero-bytes segments, the pipeline's slices) lease it from a process-wide pool (include/giddy/stream_scratch.hpp):
 * a block goes back to the stream (and host thread) that used it last, which may have it again at once, in stream order.  Without
 * scratch the wrappers take their scratch-free paths — slower for such columns, never wrong. */
int          giddy_reserve_scratch(void* hip_stream, uint64_t bytes);   /* a block of >= bytes is there for this stream afterwards: launches on it do not allocate */
int          giddy_release_scratch(void* hip_stream);                   /* call before destroying a stream: its blocks return to the pool behind an event */
uint64_t     giddy_trim(void);                                          /* frees every block not in use; WAITS for the work that used them; bytes freed */
uint64_t     giddy_scratch_held_bytes(void);
/* What a stream that is being captured into a HIP graph gets: 0 (default) nothing — scratch-free paths inside graphs;
 * 1 hipMallocAsync / hipFreeAsync, which become nodes of the graph. */
int          giddy_set_capture_scratch_policy(int policy);

/* ---- a column cut into shards (GPUs of a node) or chunks (a host-to-device pipeline) ---------------------
 * The reference has no multi-GPU or transfer layer (README.md:15-24 states the purpose — move compressed, expand on the
 * device — and leaves the moving to the caller); these entry points are this library's own, over include/giddy/sharding.hpp
 * (the same cuts as libgiddy_amd/sharding.py) and libgiddy_amd/csrc/wrappers/sharding.hip. */
typedef enum giddy_scheme {
	GIDDY_SCHEME_ELEMENTWISE = 1,                     /* Dictionary, fixed-width discard-zero-bytes */
	GIDDY_SCHEME_FRAME_OF_REFERENCE,
	GIDDY_SCHEME_DELTA,
	GIDDY_SCHEME_DISCARD_ZERO_BYTES_VARIABLE,
	GIDDY_SCHEME_RUN_LENGTH_ENCODING,
	GIDDY_SCHEME_RUN_POSITION_ENCODING_RELATIVE,
	GIDDY_SCHEME_RUN_POSITION_ENCODING_ABSOLUTE
} giddy_scheme_t;

/* what the cuts need to know of a column */
typedef struct giddy_column_shape {
	uint64_t     length;                  /* decoded elements */
	uint64_t     segment_length;          /* elements per anchored segment (ignored: elementwise) */
	uint32_t     output_itemsize;         /* bytes per decoded element */
	uint32_t     bits_per_element_size;   /* variable-width discard-zero-bytes */
	uint32_t     size_container_size;     /* ... bytes per container of size fields */
	uint32_t     anchor_itemsize;         /* bytes per position anchor (variable-rate schemes) */
	uint64_t     num_anchors;             /* entries of position_anchors */
	uint64_t     num_units;               /* compressed_data bytes (discard-zero-bytes) / element runs (run encodings) */
	const void*  position_anchors;        /* the column's anchors: two of them are read */
	int32_t      anchors_on_device;       /* 1: position_anchors is device memory (two synchronous scalar copies) */
	int32_t      reserved;
} giddy_column_shape_t;

/* cuda::sharding::shard_t: slices [first, end) per array, what to subtract from the shard's anchors / absolute positions */
typedef struct giddy_shard {
	uint64_t  first_segment, end_segment;                 /* per-segment arrays: anchors, anchor values, model records */
	uint64_t  first_element, end_element;                 /* per-element arrays: compressed_input; the output */
	uint64_t  first_unit, end_unit;                       /* compressed_data bytes / runs */
	uint64_t  first_size_container, end_size_container;   /* packed_element_sizes */
	uint64_t  anchor_bias, position_bias;
} giddy_shard_t;

int          giddy_shard(int scheme, const giddy_column_shape_t* shape, uint64_t rank, uint64_t world_size, giddy_shard_t* shard);

/* out[i] = in[i] - bias (wrapping, `itemsize` 1, 2, 4 or 8 bytes per entry) on `hip_stream`: an anchor slice — or a slice of absolute
 * run positions — rebased to its shard where it lies, on the device; out == in is allowed */
int          giddy_enqueue_rebase(void* hip_stream, void* out, const void* in, uint64_t count, uint32_t itemsize, uint64_t bias);

/* One process, every device of a node (examples/src/sharded_decode_on_all_devices.hip): the DECODED shards — shard i: `bytes[i]` bytes at
 * `shards[i]` on device `devices[i]`, written by work enqueued on `hip_streams[i]` — into one contiguous buffer at `destination` on
 * `root_device`, shard after shard.  Each shard's copy is enqueued on its own stream (it crosses its xGMI link once, behind its decode);
 * `root_stream` (a stream of the root device) then waits for all of them: work enqueued on it afterwards sees the whole column.  Peer access
 * between each shard's device and the root is checked and enabled once per pair (hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess); where they
 * cannot reach each other directly the copies are staged through the host by the runtime — slower, said once on stderr — unless
 * GIDDY_REQUIRE_PEER_ACCESS=1 is in the environment, which makes it an error.  The
 * counterpart of libgiddy_amd.distributed.gather_decoded (one process per GPU, RCCL) for callers who drive all devices themselves. */
int          giddy_enqueue_gather_on_node(void* destination, int root_device, void* root_stream, const void* const* shards, const int* devices,
                                          void* const* hip_streams, const uint64_t* bytes, size_t num_shards);

/* The same column on the root, the other way round — move COMPRESSED, expand at the destination (the reference's premise, README.md:15-24;
 * the one-process counterpart of libgiddy_amd.distributed.gather_compressed): shard i's compressed arrays lie on shards[i].device; each is
 * copied ONCE across its link into scratch of a root stream (hipMemcpyPeerAsync on root_streams[i % num_root_streams], behind what the
 * shard's own stream has enqueued so far) and decoded there, by the wrapper `key` names, to where its "decompressed" argument points ON THE
 * ROOT — shard i + 1's copy overlaps shard i's decode when there are two root streams or more.  root_streams[0] then waits for the others.
 * For discard-zero-bytes columns that is 2.9 x fewer bytes over xGMI than gathering decoded shards.  Peer access is checked and enabled
 * as for giddy_enqueue_gather_on_node (staged through the host where there is none; GIDDY_SHARD_DECODE_IN_PLACE needs it and is an error without).
 * Cut a shard finer (giddy_shard) for more overlap. */
#define GIDDY_SHARD_COPY_EVEN_IF_LOCAL 1u    /* copy the arrays into the root's scratch even when the shard already lies on the root (tests on one GPU) */
#define GIDDY_SHARD_DECODE_IN_PLACE    2u    /* no copy: the root's decoder reads the shard's arrays where they lie, over the link (peer access) — every byte crosses once, nothing is staged */
typedef struct giddy_compressed_shard {
	int32_t                  device;                    /* where the shard's compressed arrays lie */
	uint32_t                 flags;
	void*                    hip_stream;                /* a stream of THAT device: the arrays are complete once what it holds so far has run */
	const giddy_argument_t*  arguments;                 /* the wrapper's launch arguments for this shard (anchors rebased, sizes the shard's): inputs on `device`, "decompressed" on the root */
	const uint64_t*          bytes;                     /* per argument: bytes behind a pointer that is to travel; 0 for scalars and for "decompressed" */
	size_t                   num_arguments;
	const giddy_argument_t*  resolution_arguments;      /* what giddy_resolve_launch_configuration takes for this shard (size_t values) */
	size_t                   num_resolution_arguments;
} giddy_compressed_shard_t;
int          giddy_enqueue_gather_compressed_on_node(const char* key, int root_device, void* const* root_streams, size_t num_root_streams,
                                                     const giddy_compressed_shard_t* shards, size_t num_shards);

/* The host-to-device pipeline: the column — every array argument of `arguments` except "decompressed" a HOST pointer (pinned, for the
 * copies to be asynchronous), "decompressed" a device pointer — is cut into `num_chunks` chunks (= shards), chunk c's slices are
 * copied to the device on hip_streams[c % num_streams], its anchors rebased there, and it is decoded by the wrapper `key` names into its
 * range of the output: the copy of chunk c + 1 overlaps the decoding of chunk c.  Scalars that state sizes ("length",
 * "num_segments", "num_anchors", "num_element_runs", "total_uncompressed_length", "uncompressed_length") are set per chunk, the
 * others and small side tables ("dictionary_entries": copied once per stream) passed on.  Nothing is synchronised: the caller waits on
 * the streams.  `shape->position_anchors` must be the HOST array here. */
int          giddy_decode_from_host(const char* key, int scheme, const giddy_column_shape_t* shape,
                 const giddy_argument_t* arguments, size_t num_arguments,
                 uint64_t num_chunks, void* const* hip_streams, size_t num_streams);

/* ---- diagnostics -------------------------------------------------------------------------- */
const char*  giddy_last_error(void);                   /* message of this thread's last failed call */
const char*  giddy_version(void);

#ifdef __cplusplus
}
#endif
#endif /* GIDDY_H_ */
