"""ctypes binding of oracle/libgiddy_oracle.so (the C restatement of the reference kernels).

TEST INFRASTRUCTURE ONLY — see oracle/giddy_oracle.c.  All functions take/return numpy arrays
on the host and follow the argument meaning of the reference kernels they restate.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GIDDY_ORACLE_LIBRARY_PATH selects another build of the same restatement (the sanitizer build, tests/test_sanitizers.py)
_SO = os.environ.get("GIDDY_ORACLE_LIBRARY_PATH") or os.path.join(_HERE, "libgiddy_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "giddy_oracle.c")
    if os.environ.get("GIDDY_ORACLE_LIBRARY_PATH"):
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libgiddy_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_incidence_bitmaps.restype = ctypes.c_int64
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _u64(x):
    return ctypes.c_uint64(int(x))


def max_threads():
    return int(lib().oracle_max_threads())


MODEL_KINDS = {"constant": 0, "linear": 1}


def frame_of_reference(compressed, coefficients, length, segment_length, out_dtype, model="constant",
                       model_index_size=4, threads=1, out=None):
    if out is None:
        out = np.zeros(length, dtype=out_dtype)
    coefficients = np.ascontiguousarray(coefficients)
    rc = lib().oracle_frame_of_reference(_p(out), out.itemsize, _p(compressed), compressed.itemsize,
                                         _p(coefficients), coefficients.itemsize, MODEL_KINDS[model],
                                         _u64(length), _u64(segment_length), model_index_size, threads)
    assert rc == 0, rc
    return out


def model(coefficients, length, out_dtype, model="linear", model_index_size=4, threads=1):
    out = np.zeros(length, dtype=out_dtype)
    coefficients = np.ascontiguousarray(coefficients)
    rc = lib().oracle_model(_p(out), out.itemsize, _p(coefficients), coefficients.itemsize,
                            MODEL_KINDS[model], _u64(length), model_index_size, threads)
    assert rc == 0, rc
    return out


def dictionary(keys, dictionary_entries, threads=1, out=None):
    if out is None:
        out = np.zeros(len(keys), dtype=dictionary_entries.dtype)
    rc = lib().oracle_dictionary(_p(out), out.itemsize, _p(keys), keys.itemsize, _p(dictionary_entries),
                                 _u64(len(keys)), threads)
    assert rc == 0, rc
    return out


def delta(compressed, anchors, segment_length, threads=1, out=None):
    if out is None:
        out = np.zeros(len(compressed), dtype=anchors.dtype)
    rc = lib().oracle_delta(_p(out), out.itemsize, _p(compressed), compressed.itemsize, _p(anchors),
                            _u64(len(compressed)), _u64(segment_length), threads)
    assert rc == 0, rc
    return out


def dzb_fixed(compressed, out_dtype, threads=1):
    out = np.zeros(len(compressed), dtype=out_dtype)
    rc = lib().oracle_dzb_fixed(_p(out), out.itemsize, _p(compressed), compressed.itemsize,
                                _u64(len(compressed)), threads)
    assert rc == 0, rc
    return out


def dzb_variable(compressed_data, packed_sizes, anchors, segment_length, length, min_size, bits, out_dtype,
                 threads=1, out=None):
    if out is None:
        out = np.zeros(length, dtype=out_dtype)
    rc = lib().oracle_dzb_variable(_p(out), out.itemsize, _p(compressed_data), _p(packed_sizes),
                                   packed_sizes.itemsize, _p(anchors), anchors.itemsize, _u64(segment_length),
                                   _u64(length), ctypes.c_uint(min_size), ctypes.c_uint(bits), threads)
    assert rc == 0, rc
    return out


def incidence_bitmaps(bitmaps, dictionary_entries, bitmap_length, num_bitmaps, alignment, out_dtype,
                      use_dictionary=True, threads=1, fill=0):
    out = np.full(32 * bitmap_length, fill, dtype=out_dtype)
    uncovered = lib().oracle_incidence_bitmaps(_p(out), out.itemsize, _p(bitmaps), _p(dictionary_entries),
                                               _u64(bitmap_length), ctypes.c_uint32(num_bitmaps),
                                               ctypes.c_uint32(alignment), int(use_dictionary), threads)
    return out, int(uncovered)


def run_length_encoding(run_data, run_lengths, anchors, intra_offsets, segment_length, length, threads=1,
                        out=None):
    if out is None:
        out = np.zeros(length, dtype=run_data.dtype)
    rc = lib().oracle_run_length_encoding(_p(out), out.itemsize, _p(run_data), _p(run_lengths),
                                          run_lengths.itemsize, _p(anchors), anchors.itemsize,
                                          _p(intra_offsets), _u64(segment_length), _u64(len(anchors)),
                                          _u64(len(run_data)), _u64(length), threads)
    assert rc == 0, rc
    return out


def run_position_encoding(run_data, run_start_positions, anchors, segment_length, length, relative,
                          threads=1, out=None):
    if out is None:
        out = np.zeros(length, dtype=run_data.dtype)
    rc = lib().oracle_run_position_encoding(_p(out), out.itemsize, _p(run_data), _p(run_start_positions),
                                            run_start_positions.itemsize, int(bool(relative)), _p(anchors),
                                            anchors.itemsize, _u64(segment_length), _u64(len(anchors)),
                                            _u64(len(run_data)), _u64(length), threads)
    assert rc == 0, rc
    return out


def scatter(target, data, indices, threads=1):
    rc = lib().oracle_scatter(_p(target), target.itemsize, _p(data), _p(indices), indices.itemsize,
                              _u64(len(data)), threads)
    assert rc == 0, rc
    return target


def compressed_indices_scatter(target, data, run_lengths, offset_sizes, baselines, start_pos, offset_bytes,
                               anchors, anchoring_period, threads=1):
    rc = lib().oracle_compressed_indices_scatter(
        _p(target), target.itemsize, _p(data), _p(run_lengths), run_lengths.itemsize, _p(offset_sizes),
        _p(baselines), baselines.itemsize, _p(start_pos), start_pos.itemsize, _p(offset_bytes), _p(anchors),
        _u64(anchoring_period), _u64(len(run_lengths)), _u64(len(data)), threads)
    assert rc == 0, rc
    return target
