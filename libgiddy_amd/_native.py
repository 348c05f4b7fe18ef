"""Loads libgiddy_amd.so (HIP kernels + C++ wrappers + C ABI, see include/giddy.h) through ctypes.

There is deliberately no fallback: if the shared library is missing or fails to load, importing
this module raises — the product path never runs without the native extension.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GIDDY_LIBRARY_PATH selects another build of the same library (e.g. a tuning variant); there is still no fallback
LIBRARY_PATH = os.environ.get("GIDDY_LIBRARY_PATH") or os.path.join(_HERE, "libgiddy_amd.so")


class giddy_value(ctypes.Union):
    _fields_ = [("u64", ctypes.c_uint64), ("i64", ctypes.c_int64), ("pointer", ctypes.c_void_p),
                ("i32", ctypes.c_int32 * 2), ("i16", ctypes.c_int16 * 2)]


class giddy_argument(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("type", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("value", giddy_value)]


class giddy_launch_configuration(ctypes.Structure):
    _fields_ = [("grid_dimensions", ctypes.c_uint32 * 3), ("block_dimensions", ctypes.c_uint32 * 3),
                ("dynamic_shared_memory_size", ctypes.c_uint32)]


class giddy_launch_configuration_limits(ctypes.Structure):
    _fields_ = [("has_block_size", ctypes.c_int32), ("block_size", ctypes.c_uint32),
                ("has_grid_size", ctypes.c_int32), ("grid_size", ctypes.c_uint32),
                ("has_dynamic_shared_memory", ctypes.c_int32), ("dynamic_shared_memory", ctypes.c_uint32)]


# every symbol include/giddy.h declares
EXPORTED_SYMBOLS = (
    "giddy_kernel_count", "giddy_kernel_key", "giddy_can_produce", "giddy_produce", "giddy_release",
    "giddy_resolve_launch_configuration", "giddy_enqueue_launch", "giddy_device_function",
    "giddy_device_function_count", "giddy_device_function_alternate",
    "giddy_prepare", "giddy_prepared_update", "giddy_enqueue_prepared", "giddy_release_prepared",
    "giddy_can_launch_batches", "giddy_batch_create", "giddy_batch_add", "giddy_batch_clear", "giddy_batch_num_pieces", "giddy_batch_num_blocks",
    "giddy_batch_upload", "giddy_enqueue_batch", "giddy_batch_tables", "giddy_enqueue_batch_of_device_tables", "giddy_batch_release",
    "giddy_reserve_scratch", "giddy_release_scratch", "giddy_trim", "giddy_scratch_held_bytes", "giddy_set_capture_scratch_policy",
    "giddy_shard", "giddy_enqueue_rebase", "giddy_decode_from_host", "giddy_enqueue_gather_on_node", "giddy_enqueue_gather_compressed_on_node",
    "giddy_last_error", "giddy_version",
)


class giddy_batch_tables(ctypes.Structure):
    _fields_ = [("pieces", ctypes.c_void_p), ("blocks", ctypes.c_void_p), ("num_pieces", ctypes.c_uint32), ("num_blocks", ctypes.c_uint32),
                ("block_size", ctypes.c_uint32), ("dynamic_shared_memory_size", ctypes.c_uint32), ("hints", ctypes.c_uint64 * 4)]


class giddy_column_shape(ctypes.Structure):
    _fields_ = [("length", ctypes.c_uint64), ("segment_length", ctypes.c_uint64), ("output_itemsize", ctypes.c_uint32),
                ("bits_per_element_size", ctypes.c_uint32), ("size_container_size", ctypes.c_uint32), ("anchor_itemsize", ctypes.c_uint32),
                ("num_anchors", ctypes.c_uint64), ("num_units", ctypes.c_uint64), ("position_anchors", ctypes.c_void_p),
                ("anchors_on_device", ctypes.c_int32), ("reserved", ctypes.c_int32)]


class giddy_shard(ctypes.Structure):
    _fields_ = [(name, ctypes.c_uint64) for name in (
        "first_segment", "end_segment", "first_element", "end_element", "first_unit", "end_unit",
        "first_size_container", "end_size_container", "anchor_bias", "position_bias")]


class giddy_compressed_shard(ctypes.Structure):
    _fields_ = [("device", ctypes.c_int32), ("flags", ctypes.c_uint32), ("hip_stream", ctypes.c_void_p),
                ("arguments", ctypes.POINTER(giddy_argument)), ("bytes", ctypes.POINTER(ctypes.c_uint64)), ("num_arguments", ctypes.c_size_t),
                ("resolution_arguments", ctypes.POINTER(giddy_argument)), ("num_resolution_arguments", ctypes.c_size_t)]


SCHEMES = {"elementwise": 1, "frame_of_reference": 2, "delta": 3, "discard_zero_bytes_variable": 4, "run_length_encoding": 5,
           "run_position_encoding_relative": 6, "run_position_encoding_absolute": 7}


def _load():
    if not os.path.exists(LIBRARY_PATH):
        raise ImportError(
            "libgiddy_amd.so is not built (%s). Build it with `make -C libgiddy_amd/csrc -j8` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`; there is no non-native fallback." % LIBRARY_PATH)
    lib = ctypes.CDLL(LIBRARY_PATH, mode=ctypes.RTLD_GLOBAL)
    lib.giddy_kernel_count.restype = ctypes.c_int
    lib.giddy_kernel_key.restype = ctypes.c_char_p
    lib.giddy_kernel_key.argtypes = [ctypes.c_int]
    lib.giddy_can_produce.argtypes = [ctypes.c_char_p]
    lib.giddy_produce.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
    lib.giddy_release.argtypes = [ctypes.c_void_p]
    lib.giddy_release.restype = None
    lib.giddy_resolve_launch_configuration.argtypes = [
        ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(giddy_argument), ctypes.c_size_t,
        ctypes.POINTER(giddy_launch_configuration_limits), ctypes.POINTER(giddy_launch_configuration)]
    lib.giddy_enqueue_launch.argtypes = [
        ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(giddy_launch_configuration),
        ctypes.POINTER(giddy_argument), ctypes.c_size_t]
    lib.giddy_device_function.restype = ctypes.c_void_p
    lib.giddy_device_function.argtypes = [ctypes.c_void_p]
    lib.giddy_device_function_count.argtypes = [ctypes.c_void_p]
    lib.giddy_device_function_alternate.restype = ctypes.c_void_p
    lib.giddy_device_function_alternate.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.giddy_prepare.argtypes = [ctypes.c_void_p, ctypes.POINTER(giddy_argument), ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]
    lib.giddy_prepared_update.argtypes = [ctypes.c_void_p, ctypes.POINTER(giddy_argument), ctypes.c_size_t]
    lib.giddy_enqueue_prepared.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(giddy_launch_configuration)]
    lib.giddy_release_prepared.argtypes = [ctypes.c_void_p]
    lib.giddy_release_prepared.restype = None
    lib.giddy_can_launch_batches.argtypes = [ctypes.c_void_p]
    lib.giddy_batch_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p)]
    lib.giddy_batch_add.argtypes = [ctypes.c_void_p, ctypes.POINTER(giddy_argument), ctypes.c_size_t, ctypes.POINTER(giddy_argument), ctypes.c_size_t]
    lib.giddy_batch_clear.argtypes = [ctypes.c_void_p]
    lib.giddy_batch_num_pieces.argtypes = [ctypes.c_void_p]
    lib.giddy_batch_num_pieces.restype = ctypes.c_uint64
    lib.giddy_batch_num_blocks.argtypes = [ctypes.c_void_p]
    lib.giddy_batch_num_blocks.restype = ctypes.c_uint64
    lib.giddy_batch_upload.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.giddy_enqueue_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.giddy_batch_tables.argtypes = [ctypes.c_void_p, ctypes.POINTER(giddy_batch_tables)]
    lib.giddy_enqueue_batch_of_device_tables.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(giddy_batch_tables)]
    lib.giddy_batch_release.argtypes = [ctypes.c_void_p]
    lib.giddy_batch_release.restype = None
    lib.giddy_reserve_scratch.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
    lib.giddy_release_scratch.argtypes = [ctypes.c_void_p]
    lib.giddy_trim.restype = ctypes.c_uint64
    lib.giddy_scratch_held_bytes.restype = ctypes.c_uint64
    lib.giddy_set_capture_scratch_policy.argtypes = [ctypes.c_int]
    lib.giddy_enqueue_gather_on_node.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int),
                                                 ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t]
    lib.giddy_enqueue_gather_on_node.restype = ctypes.c_int
    lib.giddy_enqueue_gather_compressed_on_node.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t,
                                                            ctypes.POINTER(giddy_compressed_shard), ctypes.c_size_t]
    lib.giddy_shard.argtypes = [ctypes.c_int, ctypes.POINTER(giddy_column_shape), ctypes.c_uint64, ctypes.c_uint64, ctypes.POINTER(giddy_shard)]
    lib.giddy_enqueue_rebase.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64]
    lib.giddy_decode_from_host.argtypes = [ctypes.c_char_p, ctypes.c_int, ctypes.POINTER(giddy_column_shape), ctypes.POINTER(giddy_argument), ctypes.c_size_t,
                                           ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    lib.giddy_last_error.restype = ctypes.c_char_p
    lib.giddy_version.restype = ctypes.c_char_p
    return lib


lib = _load()
