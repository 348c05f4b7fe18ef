"""TEST INFRASTRUCTURE ONLY — numpy statement of the reference's un-anchored scan kernels
(src/kernels/reduction/scan.cuh): what `reduce_segments` (:125-186) writes per segment, and what
`scan_using_segment_reductions` (:406-460) / `scan_single_segment` (:519-540) leave in the result array —
a prefix reduction of the pre-transformed input over the WHOLE array (phase 2 seeds every segment with the
reduction of the preceding segments' results, :420-443), inclusive or exclusive (:17, `inclusivity_t`).
Integer arithmetic wraps in the result type, as the device's does.  Parity unpinned by reference-held vectors:
the reference has no tests for these kernels."""
import numpy as np

_ACCUMULATE = {"plus": np.add, "maximum": np.maximum, "minimum": np.minimum, "bit_or": np.bitwise_or,
               "bit_and": np.bitwise_and, "bit_xor": np.bitwise_xor, "multiplies": np.multiply}


def neutral_value(op, dtype):
    """functors.hpp:112-208"""
    dtype = np.dtype(dtype)
    if op in ("plus", "bit_or", "bit_xor"):
        return dtype.type(0)
    if op == "multiplies":
        return dtype.type(1)
    if op == "bit_and":
        return dtype.type(~dtype.type(0))
    info = np.iinfo(dtype) if dtype.kind in "iu" else np.finfo(dtype)
    return dtype.type(info.max if op == "minimum" else info.min)


def pretransform(data, result_dtype, pre="identity"):
    x = np.asarray(data).astype(result_dtype)   # identity<result_type> / cast<In, result_type>: a conversion
    if pre == "absolute":
        with np.errstate(over="ignore"):
            x = np.where(x < 0, -x, x).astype(result_dtype)
    return x


def reduce_segments(data, segment_length, op, result_dtype, pre="identity"):
    x = pretransform(data, result_dtype, pre)
    n = len(x)
    num_segments = -(-n // segment_length)
    out = np.empty(num_segments, dtype=result_dtype)
    with np.errstate(over="ignore"):
        for s in range(num_segments):
            out[s] = _ACCUMULATE[op].reduce(x[s * segment_length:(s + 1) * segment_length], dtype=result_dtype)
    return out


def scan(data, op, result_dtype, inclusive=True, pre="identity"):
    x = pretransform(data, result_dtype, pre)
    with np.errstate(over="ignore"):
        inc = _ACCUMULATE[op].accumulate(x, dtype=result_dtype) if len(x) else x
    if inclusive:
        return inc
    out = np.empty_like(inc)
    if len(out):
        out[0] = neutral_value(op, result_dtype)
        out[1:] = inc[:-1]
    return out
