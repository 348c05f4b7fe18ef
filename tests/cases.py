"""Seeded synthetic columns + their compressed forms, shared by the CPU (oracle) and GPU (parity) tests.

Each builder returns (fields, expected_column) where `fields` is the dict produced by
libgiddy_amd.formats (argument names of the reference wrappers) and `expected_column` is the
uncompressed column the writer started from — i.e. the round-trip truth.
"""
import numpy as np

from libgiddy_amd import formats


def rng(seed):
    return np.random.default_rng(seed)


def for_case(n, segment_length, out_dtype=np.int32, comp_dtype=np.int16, model="constant", seed=1):
    r = rng(seed)
    info = np.iinfo(comp_dtype)
    nseg = -(-n // segment_length)
    comp = r.integers(info.min, info.max, size=n, endpoint=True).astype(comp_dtype)
    dim = 1 if model == "constant" else 2
    coefs = r.integers(-1000000, 1000000, size=(nseg, dim)).astype(np.int32)
    if model == "linear":
        coefs[:, 1] = r.integers(-50, 50, size=nseg)
    return {"compressed_input": comp, "segment_model_coefficients": coefs, "length": n,
            "segment_length": segment_length}


def dictionary_case(n, num_entries=256, value_dtype=np.uint32, key_dtype=np.uint8, seed=2, zipf=False):
    r = rng(seed)
    info = np.iinfo(value_dtype)
    entries = r.integers(0, info.max, size=num_entries, endpoint=True, dtype=np.uint64).astype(value_dtype)
    if zipf:
        keys = np.minimum(r.zipf(1.2, size=n) - 1, num_entries - 1)
    else:
        keys = r.integers(0, num_entries, size=n)
    return {"compressed_input": keys.astype(key_dtype), "dictionary_entries": entries, "length": n,
            "num_dictionary_entries": num_entries}


def delta_case(n, segment_length, out_dtype=np.uint64, comp_dtype=np.uint16, seed=3, base=1 << 40):
    r = rng(seed)
    out_dtype = np.dtype(out_dtype)
    d = r.integers(0, np.iinfo(comp_dtype).max, size=n, endpoint=True).astype(comp_dtype)
    base = base % (1 << (8 * out_dtype.itemsize))
    total = (np.cumsum(d.astype(np.uint64), dtype=np.uint64) + np.uint64(base)).astype(out_dtype)
    before = np.concatenate((np.array([base], dtype=out_dtype), total[:-1]))
    anchors = before[np.arange(0, n, segment_length)] if n else np.zeros(0, out_dtype)
    return ({"compressed_input": d, "anchor_values": np.ascontiguousarray(anchors), "length": n,
             "segment_length": segment_length}, total)


def patch_case(n, num_patches, value_dtype=np.uint64, index_size=4, seed=4):
    r = rng(seed)
    idx = np.sort(r.choice(n, size=num_patches, replace=False)).astype(formats.uint_t(index_size))
    vals = r.integers(0, np.iinfo(value_dtype).max, size=num_patches, endpoint=True, dtype=np.uint64).astype(value_dtype)
    return {"indices": idx, "data": vals, "data_length": num_patches}


def runs_column(n, mean_run=32, value_dtype=np.uint32, seed=6, max_run=65535):
    """i.i.d. geometric run lengths, adjacent runs distinct."""
    r = rng(seed)
    est = int(n / mean_run * 1.3) + 16
    lengths = np.clip(r.geometric(1.0 / mean_run, size=est), 1, max_run)
    while lengths.sum() < n:
        lengths = np.concatenate((lengths, np.clip(r.geometric(1.0 / mean_run, size=est), 1, max_run)))
    cut = np.searchsorted(np.cumsum(lengths), n, side="left") + 1
    lengths = lengths[:cut].copy()
    lengths[-1] -= lengths.sum() - n
    info = np.iinfo(value_dtype)
    vals = r.integers(0, info.max, size=len(lengths), endpoint=True, dtype=np.uint64).astype(value_dtype)
    same = np.flatnonzero(vals[1:] == vals[:-1]) + 1
    vals[same] = vals[same] + np.array(1, dtype=value_dtype)
    return np.repeat(vals, lengths)[:n]


def dzbv_column(n, out_dtype=np.uint64, min_size=1, bits=2, seed=8):
    """Uniform element sizes in [min_size, min(min_size + 2^bits - 1, sizeof)], value with exactly
    that many significant bytes."""
    r = rng(seed)
    usize = np.dtype(out_dtype).itemsize
    hi = min(min_size + (1 << bits) - 1, usize)
    sizes = r.integers(max(min_size, 0), hi, size=n, endpoint=True)
    vals = r.integers(0, np.iinfo(np.uint64).max, size=n, endpoint=True, dtype=np.uint64)
    mask = np.where(sizes >= 8, np.uint64(0xFFFFFFFFFFFFFFFF),
                    (np.uint64(1) << (sizes.astype(np.uint64) * np.uint64(8))) - np.uint64(1))
    vals &= mask
    top = np.where(sizes > 0, np.uint64(1) << ((np.maximum(sizes, 1).astype(np.uint64) - np.uint64(1)) * np.uint64(8)), np.uint64(0))
    vals |= top  # make the top byte non-zero so the size is exact
    return vals.astype(out_dtype), sizes
