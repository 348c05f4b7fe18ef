"""Python mirror of the reference's type-erased, factory-provided kernel wrappers ("mode of use A",
examples/src/modes_of_use/factory_provided_type_erased_wrapper.cu:7-45) over the C ABI.

    kernel = registered.kernel_t.produceSubclass(
        "decompression::frame_of_reference::kernel_t<4u, int, short, "
        "cuda::functors::unary::parametric_model::constant<4u, int> >")
    config = kernel.resolve_launch_configuration(device, {"length": size_t(n), "segment_length": size_t(L)})
    kernel.enqueue_launch(stream, config, {"decompressed": pointer(out, "i32"), ...})

Same method names, same argument-map keys, same error behaviour as cuda::registered::kernel_t
(src/kernel_wrappers/registered_wrapper.h:28-39, src/cuda/kernel_wrapper.cuh:53-128): C++
exceptions surface as the Python exceptions below.
"""
import ctypes

import numpy as np

from . import _native
from ._native import giddy_argument, giddy_launch_configuration, giddy_launch_configuration_limits

NOMINAL_MI355X = -1  # GIDDY_NOMINAL_MI355X_DEVICE: resolve without touching a device
CURRENT_DEVICE = -2  # GIDDY_CURRENT_DEVICE: the calling thread's current HIP device

# giddy_type_t
_TAGS = {name: i + 1 for i, name in enumerate([
    "bool", "u8", "u16", "u32", "u64", "i8", "i16", "i32", "i64",
    "ptr_u8", "ptr_u16", "ptr_u32", "ptr_u64", "ptr_i8", "ptr_i16", "ptr_i32", "ptr_i64",
    "cptr_u8", "cptr_u16", "cptr_u32", "cptr_u64", "cptr_i8", "cptr_i16", "cptr_i32", "cptr_i64",
    "coef_i16x1", "coef_i16x2", "coef_i32x1", "coef_i32x2",
    "cptr_coef_i16x1", "cptr_coef_i16x2", "cptr_coef_i32x1", "cptr_coef_i32x2"])}


class InvalidArgument(ValueError):
    """std::invalid_argument"""


class LogicError(RuntimeError):
    """std::logic_error"""


class OutOfRange(KeyError):
    """std::out_of_range — a required argument-map key is missing"""


class BadAnyCast(TypeError):
    """bad_any_cast — an argument holds a different C++ type than the wrapper casts to"""


class HipRuntimeError(RuntimeError):
    """the HIP runtime reported an error"""


_ERRORS = {-1: InvalidArgument, -2: LogicError, -3: OutOfRange, -4: BadAnyCast, -5: HipRuntimeError}


def _check(status):
    if status != 0:
        message = (_native.lib.giddy_last_error() or b"").decode()
        raise _ERRORS.get(status, RuntimeError)(message)


class Argument:
    """A value together with the exact C++ type the wrapper will any_cast it to."""
    __slots__ = ("tag", "value", "keepalive")

    def __init__(self, tag, value, keepalive=None):
        self.tag, self.value, self.keepalive = tag, value, keepalive

    def __repr__(self):
        return "Argument(%s, %r)" % (self.tag, self.value)


def _scalar(tag):
    def make(value):
        return Argument(tag, int(value))
    make.__name__ = tag
    return make


boolean = _scalar("bool")
uint8, uint16, uint32, uint64 = _scalar("u8"), _scalar("u16"), _scalar("u32"), _scalar("u64")
int8, int16, int32, int64 = _scalar("i8"), _scalar("i16"), _scalar("i32"), _scalar("i64")
size_t = uint64                       # size_t is unsigned long, the same type as uint64_t here
serialization_factor_t = uint16       # unsigned short


def uint_t(size):
    """util::uint_t<size> by value"""
    return {1: uint8, 2: uint16, 4: uint32, 8: uint64}[size]


def size_type_by_index_size(size):
    return uint32 if size <= 4 else uint64


def _address(buffer):
    if isinstance(buffer, int):
        return buffer, None
    if hasattr(buffer, "data_ptr"):          # torch tensor (device memory)
        return buffer.data_ptr(), buffer
    if isinstance(buffer, np.ndarray):       # host memory: only meaningful for error-path tests
        return buffer.ctypes.data, buffer
    raise TypeError("cannot take the address of %r" % type(buffer))


def _dtype_code(dtype):
    dtype = np.dtype(dtype) if not isinstance(dtype, str) or dtype not in _TAGS else dtype
    if isinstance(dtype, str):
        return dtype
    return ("u" if dtype.kind == "u" else "i") + str(8 * dtype.itemsize)


def pointer(buffer, dtype):
    """T* (an output buffer); dtype names T: numpy dtype or 'u8'...'i64'"""
    address, keep = _address(buffer)
    return Argument("ptr_" + _dtype_code(dtype), address, keep)


def const_pointer(buffer, dtype):
    """const T* (an input buffer)"""
    address, keep = _address(buffer)
    return Argument("cptr_" + _dtype_code(dtype), address, keep)


def coefficients(values, dtype):
    """cuda::array<T, N> by value (Model decompression's model_coefficients)"""
    values = [int(v) for v in np.asarray(values).reshape(-1)]
    return Argument("coef_%sx%d" % (_dtype_code(dtype), len(values)), values)


def const_coefficients_pointer(buffer, dtype, dimension):
    """const cuda::array<T, N>* (Frame-of-Reference's segment_model_coefficients)"""
    address, keep = _address(buffer)
    return Argument("cptr_coef_%sx%d" % (_dtype_code(dtype), dimension), address, keep)


def _marshal(arguments):
    array = (giddy_argument * max(len(arguments), 1))()
    keep = []
    for slot, (name, argument) in zip(array, arguments.items()):
        if not isinstance(argument, Argument):
            raise TypeError("argument %r must be built with libgiddy_amd.registered's typed constructors "
                            "(size_t, uint_t(n), pointer, const_pointer, ...): the wrappers cast to exact C++ types" % name)
        encoded = name.encode()
        keep.append(encoded)
        slot.name = encoded
        slot.type = _TAGS[argument.tag]
        if argument.tag.startswith("coef_"):
            field = slot.value.i32 if "i32" in argument.tag else slot.value.i16
            for i, v in enumerate(argument.value):
                field[i] = v
        elif argument.tag.startswith(("ptr_", "cptr_")):
            slot.value.pointer = argument.value
        elif argument.tag.startswith("i"):
            slot.value.i64 = argument.value
        else:
            slot.value.u64 = argument.value
    return array, keep


class launch_configuration_t:
    """cuda::launch_configuration_t"""

    def __init__(self, grid_dimensions=(1, 1, 1), block_dimensions=(1, 1, 1), dynamic_shared_memory_size=0):
        self.grid_dimensions = tuple(grid_dimensions)
        self.block_dimensions = tuple(block_dimensions)
        self.dynamic_shared_memory_size = dynamic_shared_memory_size

    def _c(self):
        c = giddy_launch_configuration()
        c.grid_dimensions[:] = self.grid_dimensions
        c.block_dimensions[:] = self.block_dimensions
        c.dynamic_shared_memory_size = self.dynamic_shared_memory_size
        return c

    def __repr__(self):
        return "launch_configuration_t(grid=%s, block=%s, dynamic_shared_memory=%d)" % (
            self.grid_dimensions, self.block_dimensions, self.dynamic_shared_memory_size)


class launch_configuration_limits_t:
    """cuda::launch_configuration_limits_t (src/cuda/kernel_wrapper.cuh:20-33); None = unset"""

    def __init__(self, block_size=None, grid_size=None, dynamic_shared_memory=None):
        self.block_size, self.grid_size, self.dynamic_shared_memory = block_size, grid_size, dynamic_shared_memory

    def _c(self):
        c = giddy_launch_configuration_limits()
        for field in ("block_size", "grid_size", "dynamic_shared_memory"):
            value = getattr(self, field)
            setattr(c, "has_" + field, int(value is not None))
            setattr(c, field, int(value or 0))
        return c


class prepared_launch:
    """giddy_prepared_t: a wrapper with an argument map that is kept between launches; update() rewrites entries in place"""

    def __init__(self, kernel, arguments):
        self._kernel = kernel   # (keeps the wrapper alive)
        self._handle = ctypes.c_void_p()
        array, keep = _marshal(arguments)
        _check(_native.lib.giddy_prepare(kernel._handle, array, len(arguments), ctypes.byref(self._handle)))

    def update(self, arguments):
        array, keep = _marshal(arguments)
        _check(_native.lib.giddy_prepared_update(self._handle, array, len(arguments)))

    def enqueue(self, stream, launch_config):
        c_config = launch_config._c() if not isinstance(launch_config, giddy_launch_configuration) else launch_config
        _check(_native.lib.giddy_enqueue_prepared(self._handle, ctypes.c_void_p(int(stream) if stream else None), ctypes.byref(c_config)))

    def __del__(self):
        handle, self._handle = getattr(self, "_handle", None), None
        if handle and _native is not None and getattr(_native, "lib", None) is not None:
            _native.lib.giddy_release_prepared(handle)


class batch_t:
    """giddy_batch_t (include/giddy/batch.hpp): MANY small columns decoded by ONE launch.

        batch = kernel.batch(device)                                  # wrappers for which kernel.launches_batches()
        for each column: batch.add(launch_arguments, resolution_arguments)   # what enqueue_launch / resolve_launch_configuration take for it; host only
        batch.enqueue(stream)                                         # the tables to the device (first time / after a change) and ONE launch
    Keep it alive until the launches that read its tables have run."""

    def __init__(self, kernel, device=CURRENT_DEVICE):
        self._kernel = kernel
        self._handle = ctypes.c_void_p()
        self._keep = []   # the buffers behind the pieces' pointers
        _check(_native.lib.giddy_batch_create(kernel._handle, int(device), ctypes.byref(self._handle)))

    def add(self, launch_arguments, resolution_arguments):
        launch, keep_launch = _marshal(launch_arguments)
        resolve, keep_resolve = _marshal(resolution_arguments)
        _check(_native.lib.giddy_batch_add(self._handle, launch, len(launch_arguments), resolve, len(resolution_arguments)))
        self._keep.append([a.keepalive for a in launch_arguments.values() if a.keepalive is not None])

    def clear(self):
        _check(_native.lib.giddy_batch_clear(self._handle))
        self._keep = []

    @property
    def num_pieces(self):
        return int(_native.lib.giddy_batch_num_pieces(self._handle))

    @property
    def num_blocks(self):
        """workgroups of the one launch"""
        return int(_native.lib.giddy_batch_num_blocks(self._handle))

    def upload(self, stream):
        _check(_native.lib.giddy_batch_upload(self._handle, ctypes.c_void_p(int(stream) if stream else None)))

    def enqueue(self, stream):
        _check(_native.lib.giddy_enqueue_batch(self._handle, ctypes.c_void_p(int(stream) if stream else None)))

    def tables(self):
        """where the uploaded tables lie (giddy_batch_tables_t): for giddy_enqueue_batch_of_device_tables"""
        out = _native.giddy_batch_tables()
        _check(_native.lib.giddy_batch_tables(self._handle, ctypes.byref(out)))
        return out

    def __del__(self):
        handle, self._handle = getattr(self, "_handle", None), None
        if handle and _native is not None and getattr(_native, "lib", None) is not None:
            _native.lib.giddy_batch_release(handle)


def reserve_scratch(stream, nbytes):
    """cuda::stream_scratch::reserve: a block of at least `nbytes` is there for `stream` afterwards (launches on it do not allocate)"""
    _check(_native.lib.giddy_reserve_scratch(ctypes.c_void_p(int(stream) if stream else None), int(nbytes)))


def release_scratch(stream):
    """cuda::stream_scratch::forget: call before destroying a stream"""
    _check(_native.lib.giddy_release_scratch(ctypes.c_void_p(int(stream) if stream else None)))


def trim():
    """frees every scratch block that is not in use (waits for the work that used them); bytes freed"""
    return int(_native.lib.giddy_trim())


def scratch_held_bytes():
    return int(_native.lib.giddy_scratch_held_bytes())


def set_capture_scratch_policy(graph_allocation):
    _check(_native.lib.giddy_set_capture_scratch_policy(1 if graph_allocation else 0))


class kernel_t:
    """cuda::registered::kernel_t"""

    def __init__(self, handle, key):
        self._handle, self.key = handle, key

    # ---- factory (util::mixins::FactoryProducible) ------------------------------------------
    @staticmethod
    def produceSubclass(key):
        handle = ctypes.c_void_p()
        _check(_native.lib.giddy_produce(key.encode(), ctypes.byref(handle)))
        return kernel_t(handle, key)

    @staticmethod
    def canProduce(key):
        return bool(_native.lib.giddy_can_produce(key.encode()))

    @staticmethod
    def registered_keys():
        """keys of getSubclassFactory().getInsantiators(), sorted"""
        return [_native.lib.giddy_kernel_key(i).decode() for i in range(_native.lib.giddy_kernel_count())]

    # ---- cuda::kernel_t -------------------------------------------------------------------------
    def resolve_launch_configuration(self, device, extra_arguments, limits=None):
        """`device`: a HIP device ordinal, CURRENT_DEVICE, or NOMINAL_MI355X to resolve on the host alone"""
        array, keep = _marshal(extra_arguments)
        out = giddy_launch_configuration()
        c_limits = limits._c() if limits is not None else None
        _check(_native.lib.giddy_resolve_launch_configuration(
            self._handle, int(device), array, len(extra_arguments),
            ctypes.byref(c_limits) if c_limits is not None else None, ctypes.byref(out)))
        return launch_configuration_t(tuple(out.grid_dimensions), tuple(out.block_dimensions),
                                      out.dynamic_shared_memory_size)

    def enqueue_launch(self, stream, launch_config, arguments):
        """`stream`: a hipStream_t as an integer (torch: torch.cuda.current_stream().cuda_stream); 0 = default"""
        array, keep = _marshal(arguments)
        c_config = launch_config._c()
        _check(_native.lib.giddy_enqueue_launch(
            self._handle, ctypes.c_void_p(int(stream) if stream else None), ctypes.byref(c_config),
            array, len(arguments)))

    def get_device_function(self):
        return _native.lib.giddy_device_function(self._handle)

    def device_functions(self):
        """the kernels the wrapper may launch with the caller's configuration: [0] is get_device_function(), the kernel of the
        reference's signature; the others are the builds its typed launch picks among (cuda::device_function_t's alternates)"""
        return [_native.lib.giddy_device_function_alternate(self._handle, i) for i in range(_native.lib.giddy_device_function_count(self._handle))]

    def launches_batches(self):
        return bool(_native.lib.giddy_can_launch_batches(self._handle))

    def batch(self, device=CURRENT_DEVICE):
        """an empty batch of this wrapper's columns (giddy_batch_create)"""
        return batch_t(self, device)

    def enqueue_launch_of_batch(self, stream, tables):
        """one launch over tables that lie on the device already (a giddy_batch_tables: batch_t.tables(), or written by the caller's own kernel)"""
        _check(_native.lib.giddy_enqueue_batch_of_device_tables(self._handle, ctypes.c_void_p(int(stream) if stream else None), ctypes.byref(tables)))

    def prepare(self, arguments):
        """a launch whose argument map is built once and kept (giddy_prepare): for loops over a column's chunks"""
        return prepared_launch(self, arguments)

    def __del__(self):
        handle, self._handle = getattr(self, "_handle", None), None
        if handle and _native is not None and getattr(_native, "lib", None) is not None:   # (module globals are gone at interpreter exit)
            _native.lib.giddy_release(handle)
