"""The reference's three modes of use for one device-resident column (tests/cpp/modes_shim.hip behind ctypes).

    mode "A"   the factory's type-erased wrapper through the C ABI (libgiddy_amd.schemes — what every other test and bench uses)
    mode "B"   the wrapper instantiated in C++, cuda::kernel::enqueue_launch(kernel, stream, config, typed parameters...)
    mode "C"   the kernel template `decompress` launched directly (what the reference's direct_use_of_kernel.cu does)
    mode "C+"  <scheme>::launch<...>(): the one-line replacement for C's launch that reaches the wrapper's choice of kernels
    mode "B*"  cuda::kernel::enqueue_launch through a reference to the base type

A `column` here is (scheme name, dict of device tensors and scalars, output tensor); `launcher(column, mode)` returns a callable
run(stream) that enqueues one decode.  Used by tests/test_gpu_modes.py (bits) and tools/modes_bench.py (rates)."""
import ctypes
import os
import subprocess

import torch

from libgiddy_amd import schemes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tests", "cpp", "bin", "libmodes_shim.so")
MODE_NUMBERS = {"B": 1, "C": 2, "C+": 3, "B*": 4}
_shim = None


def build_shim():
    lib = os.path.join(ROOT, "libgiddy_amd")
    os.makedirs(os.path.dirname(SHIM), exist_ok=True)
    subprocess.check_call(["hipcc", "-std=c++17", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "include", "reference_paths"), os.path.join(ROOT, "tests", "cpp", "modes_shim.hip"),
                           "-o", SHIM, "-L" + lib, "-lgiddy_amd", "-Wl,-rpath," + lib])


def shim(build_if_missing=True):
    """`build_if_missing=False`: a process that has initialised the GPU (bench.py, anything under rocprofv3) never starts a compiler — it reports the shim as absent
    (build it beforehand: __graft_entry__.build_cpp_test_tools())"""
    global _shim
    if _shim is None:
        if not os.path.exists(SHIM):
            if not build_if_missing:
                raise FileNotFoundError("tests/cpp/bin/libmodes_shim.so is not built (python -c 'import __graft_entry__ as g; g.build_cpp_test_tools()')")
            build_shim()
        _shim = ctypes.CDLL(SHIM, mode=ctypes.RTLD_GLOBAL)
        _shim.modes_last_error.restype = ctypes.c_char_p
    return _shim


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _u64(v):
    return ctypes.c_uint64(int(v))


def _check(status):
    if status != 0:
        raise RuntimeError(shim().modes_last_error().decode())


# per scheme: (shim name, arguments of modes_launch_* after (mode, stream, config) from the column's fields, mode-A call)
def _delta(name):
    uncompressed_size, compressed_size = int(name.split("_")[2]), int(name.split("_")[3])

    def arguments(out, f):
        return [_p(out), _p(f["compressed_input"]), _p(f["anchor_values"]), _u64(f["length"]), _u64(f["segment_length"])]

    def mode_a(out, f, stream, config):
        return schemes.delta(out, f["compressed_input"], f["anchor_values"], f["length"], f["segment_length"],
                             uncompressed_size=uncompressed_size, compressed_size=compressed_size, stream=stream, launch_config=config)
    return name, arguments, mode_a


def _rle():
    def arguments(out, f):
        return [_p(out), _p(f["run_data"]), _p(f["run_lengths"]), _p(f["position_anchors"]), _p(f["intra_run_anchor_offsets"]),
                _u64(f["segment_length"]), _u64(f["num_segments"]), _u64(f["num_element_runs"]), _u64(f["length"])]

    def mode_a(out, f, stream, config):
        return schemes.run_length_encoding(out, f["run_data"], f["run_lengths"], f["position_anchors"], f["intra_run_anchor_offsets"], f["segment_length"],
                                           f["num_segments"], f["num_element_runs"], f["length"], stream=stream, launch_config=config)
    return "rle_4_4_2", arguments, mode_a


def _rpe(relative):
    def arguments(out, f):
        return [_p(out), _p(f["run_data"]), _p(f["run_start_positions"]), _p(f["position_anchors"]),
                _u64(f["segment_length"]), _u64(f["num_anchors"]), _u64(f["num_element_runs"]), _u64(f["length"])]

    def mode_a(out, f, stream, config):
        return schemes.run_position_encoding(out, f["run_data"], f["run_start_positions"], f["position_anchors"], f["segment_length"], f["num_anchors"],
                                             f["num_element_runs"], f["length"], position_size=2 if relative else 4, relative=relative,
                                             stream=stream, launch_config=config)
    return ("rpe_4_4_2_relative" if relative else "rpe_4_4_4_absolute"), arguments, mode_a


def _dzbv():
    def arguments(out, f):
        return [_p(out), _p(f["compressed_data"]), _p(f["packed_element_sizes"]), _p(f["position_anchors"]), _u64(f["segment_length"]), _u64(f["length"]),
                ctypes.c_uint32(int(f["min_represented_element_size"])), ctypes.c_uint32(int(f["bits_per_element_size"]))]

    def mode_a(out, f, stream, config):
        return schemes.discard_zero_bytes_variable(out, f["compressed_data"], f["packed_element_sizes"], f["position_anchors"], f["segment_length"],
                                                   f["length"], f["min_represented_element_size"], f["bits_per_element_size"], stream=stream, launch_config=config)
    return "dzbv_8_8_4", arguments, mode_a


SCHEMES = {"delta<4,8,2>": _delta("delta_4_8_2"), "delta<4,4,1>": _delta("delta_4_4_1"), "rle<4,4,2>": _rle(),
           "rpe<4,4,2,true>": _rpe(True), "rpe<4,4,4,false>": _rpe(False), "dzbv<8,8,4>": _dzbv()}


def launcher(scheme, fields, out, mode):
    """run(stream) enqueuing one decode of the column through `mode`; the launch configuration is resolved once, here, the way that mode's
    reference example resolves it"""
    name, arguments, mode_a = SCHEMES[scheme]
    if mode == "A":
        config = mode_a(out, fields, torch.cuda.current_stream().cuda_stream, None)   # (resolves, and launches once)
        return lambda stream: mode_a(out, fields, stream, config)
    number = MODE_NUMBERS[mode]
    config = (ctypes.c_uint32 * 7)()
    _check(getattr(shim(), "modes_resolve_" + name)(number, _u64(fields["length"]), _u64(fields["segment_length"]), config))
    launch = getattr(shim(), "modes_launch_" + name)
    packed = arguments(out, fields)
    return lambda stream: _check(launch(number, ctypes.c_void_p(stream), config, *packed))
