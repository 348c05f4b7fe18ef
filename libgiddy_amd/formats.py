"""Host-side format writers ("compressors") for every scheme the decompression kernels read.

The reference ships decompressors only ("no compressors", README.md:128-130); these numpy writers
define the wire layout each kernel consumes — anchor arrays, packed size words, run arrays, the
compressed-indices byte pool — exactly as the kernel documentation in the reference describes it
(file:line cited per function).  They are used to build test / benchmark inputs and by callers
that need to produce columns for the kernels.  Pure numpy, no GPU, no oracle.

Every writer returns a dict whose keys are the argument names the reference kernel wrappers use
(SURVEY.md §8b), so ``kernel.enqueue_launch(**fields)``-style plumbing reads naturally.
"""
import numpy as np

_UINT = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}
_INT = {1: np.int8, 2: np.int16, 4: np.int32, 8: np.int64}


def uint_t(n):
    """util::uint_t<N> (src/util/integer.h:194-210) as a numpy dtype."""
    return np.dtype(_UINT[n])


def int_t(n):
    return np.dtype(_INT[n])


def size_type_by_index_size(n):
    """promoted_size<uint_t<N>>: uint32 for N<=4, uint64 for N=8 (src/kernels/common.cuh:49-50)."""
    return np.dtype(np.uint32 if n <= 4 else np.uint64)


def div_rounding_up(a, b):
    return (a + b - 1) // b


# ------------------------------------------------------------------------------------------------
# Frame of reference  (src/kernels/decompression/frame_of_reference.cuh:19-33)
# ------------------------------------------------------------------------------------------------
def least_squares_slope(values):
    """round-half-even of the least-squares slope of `values` over x = 0 .. n-1, in exact integer arithmetic (0 for fewer than 2 values)"""
    n = len(values)
    if n < 2:
        return 0
    sx, sxx = n * (n - 1) // 2, (n - 1) * n * (2 * n - 1) // 6
    sy, sxy = sum(values), sum(x * v for x, v in enumerate(values))
    numerator, denominator = n * sxy - sx * sy, n * sxx - sx * sx      # denominator > 0
    quotient, remainder = divmod(numerator, denominator)                 # floor division: 0 <= remainder < denominator
    if 2 * remainder > denominator or (2 * remainder == denominator and quotient % 2 != 0):
        quotient += 1
    return quotient


def frame_of_reference(column, segment_length, compressed_dtype=np.int16, model="constant"):
    """One model record per segment + a signed per-element offset.

    constant: c0 = midpoint of the segment; linear: the least-squares slope over x = 0 .. n-1, rounded to the
    nearest integer (ties to even) in EXACT integer arithmetic — c1 = round((n Sxy - Sx Sy) / (n Sxx - Sx^2)) —
    so that the device-side writer (libgiddy_amd.compressors) reproduces it bit for bit; c0 = midpoint of the residuals
    (coefficients are int32, as in every registered instantiation,
    src/kernel_wrappers/decompression/frame_of_reference.cu:132-142).
    """
    column = np.asarray(column)
    n = len(column)
    nseg = div_rounding_up(n, segment_length)
    compressed_dtype = np.dtype(compressed_dtype)
    wide = column.astype(np.int64)
    pos = np.arange(n, dtype=np.int64) % segment_length
    seg = np.arange(n, dtype=np.int64) // segment_length
    if model == "constant":
        coefs = np.zeros((nseg, 1), dtype=np.int32)
        for s in range(nseg):
            chunk = wide[s * segment_length:(s + 1) * segment_length]
            coefs[s, 0] = np.int32((int(chunk.min()) + int(chunk.max())) // 2)
        predicted = coefs[seg, 0].astype(np.int64)
    elif model == "linear":
        coefs = np.zeros((nseg, 2), dtype=np.int32)
        for s in range(nseg):
            chunk = wide[s * segment_length:(s + 1) * segment_length]
            c1 = least_squares_slope([int(v) for v in chunk])
            resid = chunk - c1 * np.arange(len(chunk), dtype=np.int64)
            coefs[s] = (np.int32((int(resid.min()) + int(resid.max())) // 2), np.int32(c1))
        predicted = coefs[seg, 0].astype(np.int64) + coefs[seg, 1].astype(np.int64) * pos
    else:
        raise ValueError(model)
    offsets = wide - predicted
    info = np.iinfo(compressed_dtype)
    if offsets.min() < info.min or offsets.max() > info.max:
        raise ValueError("offsets do not fit the compressed type")
    return {
        "compressed_input": offsets.astype(compressed_dtype),
        "segment_model_coefficients": coefs,
        "length": n,
        "segment_length": segment_length,
    }


# ------------------------------------------------------------------------------------------------
# Dictionary  (src/kernels/decompression/dictionary.cuh:39-59)
# ------------------------------------------------------------------------------------------------
def dictionary(column, key_dtype=np.uint8):
    column = np.asarray(column)
    entries, keys = np.unique(column, return_inverse=True)
    key_dtype = np.dtype(key_dtype)
    if len(entries) > np.iinfo(key_dtype).max + 1:
        raise ValueError("too many distinct values for the key type")
    return {
        "compressed_input": keys.astype(key_dtype),
        "dictionary_entries": entries.astype(column.dtype),
        "length": len(column),
        "num_dictionary_entries": len(entries),
    }


# ------------------------------------------------------------------------------------------------
# Delta  (src/kernels/decompression/delta.cuh:5-19, :33-64)
#   out[s*L + i] = anchor[s] + sum_{j<=i} in[s*L + j];   anchor[s] is the value *before* s*L.
# ------------------------------------------------------------------------------------------------
def delta(column, segment_length, compressed_dtype=np.uint16):
    column = np.asarray(column)
    assert column.dtype.kind == "u"
    n = len(column)
    compressed_dtype = np.dtype(compressed_dtype)
    prev = np.empty_like(column)
    prev[1:] = column[:-1]
    prev[:1] = column[:1]
    d = column - prev  # modular
    starts = np.arange(0, n, segment_length)
    anchors = prev[starts].copy()
    if d.max(initial=0) > np.iinfo(compressed_dtype).max:
        raise ValueError("deltas do not fit the compressed type (column must be non-decreasing with small steps)")
    return {
        "compressed_input": d.astype(compressed_dtype),
        "anchor_values": anchors,
        "length": n,
        "segment_length": segment_length,
    }


# ------------------------------------------------------------------------------------------------
# Discard zero bytes, fixed width  (src/kernels/decompression/discard_zero_bytes_fixed.cuh:95-107)
# ------------------------------------------------------------------------------------------------
def discard_zero_bytes_fixed(column, compressed_dtype):
    column = np.asarray(column)
    compressed_dtype = np.dtype(compressed_dtype)
    if column.max(initial=0) > np.iinfo(compressed_dtype).max:
        raise ValueError("values do not fit the compressed width")
    return {"compressed_input": column.astype(compressed_dtype), "length": len(column)}


# ------------------------------------------------------------------------------------------------
# Discard zero bytes, variable width
#   (src/kernels/decompression/discard_zero_bytes_variable.cuh:23-33 size fields, :332-370 layout)
# ------------------------------------------------------------------------------------------------
def discard_zero_bytes_variable(column, segment_length, min_represented_element_size=1,
                                bits_per_element_size=2, container_dtype=np.uint32, index_size=4,
                                element_sizes=None):
    column = np.asarray(column)
    assert column.dtype.kind == "u"
    n = len(column)
    usize = column.dtype.itemsize
    container_dtype = np.dtype(container_dtype)
    if element_sizes is None:
        needed = np.zeros(n, dtype=np.int64)
        v = column.astype(np.uint64)
        for k in range(usize):
            needed[(v >> np.uint64(8 * k)) != 0] = k + 1
        element_sizes = np.maximum(needed, min_represented_element_size)
    element_sizes = np.asarray(element_sizes, dtype=np.int64)
    max_size = min_represented_element_size + (1 << bits_per_element_size) - 1
    if element_sizes.max(initial=0) > max_size or element_sizes.max(initial=0) > usize:
        raise ValueError("an element needs more bytes than the size field can express")
    per = (8 * container_dtype.itemsize) // bits_per_element_size
    ncont = div_rounding_up(n, per)
    fields = np.zeros(ncont * per, dtype=np.uint64)
    fields[:n] = (element_sizes - min_represented_element_size).astype(np.uint64)
    shifts = (np.arange(per, dtype=np.uint64) * np.uint64(bits_per_element_size))
    packed = np.bitwise_or.reduce(fields.reshape(ncont, per) << shifts[None, :], axis=1)
    offsets = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(element_sizes, out=offsets[1:])
    total = int(offsets[-1])
    raw = column.astype("<u8").view(np.uint8).reshape(n, 8) if n else np.zeros((0, 8), np.uint8)
    keep = np.arange(8)[None, :] < element_sizes[:, None]
    data = raw[keep]  # row-major: element by element, low byte first
    assert len(data) == total
    anchors = offsets[np.arange(0, n, segment_length)].astype(uint_t(index_size))
    return {
        "compressed_data": np.ascontiguousarray(data, dtype=np.uint8),
        "packed_element_sizes": packed.astype(container_dtype),
        "position_anchors": anchors,
        "segment_length": segment_length,
        "length": n,
        "min_represented_element_size": min_represented_element_size,
        "bits_per_element_size": bits_per_element_size,
    }


# ------------------------------------------------------------------------------------------------
# Incidence bitmaps  (src/kernels/decompression/incidence_bitmaps.cuh:63-131)
#   one bitmap per distinct value, each padded to `alignment` 32-bit words; length in whole words.
# ------------------------------------------------------------------------------------------------
def incidence_bitmaps(column, alignment_in_ints=2, use_dictionary=True):
    column = np.asarray(column)
    n = len(column)
    assert n % 32 == 0, "the kernel decodes whole 32-bit containers"
    entries, keys = np.unique(column, return_inverse=True)
    if not use_dictionary:
        entries = np.arange(int(column.max(initial=0)) + 1, dtype=column.dtype)
        keys = column.astype(np.int64)
    num_bitmaps = len(entries)
    assert num_bitmaps <= 256
    bitmap_length = n // 32
    aligned = div_rounding_up(bitmap_length, alignment_in_ints) * alignment_in_ints
    bitmaps = np.zeros((num_bitmaps, aligned), dtype=np.uint32)
    word = np.arange(n) // 32
    bit = (np.arange(n) % 32).astype(np.uint32)
    np.bitwise_or.at(bitmaps, (keys, word), np.uint32(1) << bit)
    return {
        "incidence_bitmaps": bitmaps.reshape(-1),
        "dictionary_entries": entries.astype(column.dtype),
        "bitmap_length": bitmap_length,
        "num_bitmaps": num_bitmaps,
    }


# ------------------------------------------------------------------------------------------------
# Run-length encoding  (src/kernels/decompression/run_length_encoding.cuh:271-326, :329-340)
# ------------------------------------------------------------------------------------------------
def _runs_of(column, max_run_length):
    n = len(column)
    if n == 0:
        return column[:0], np.zeros(0, np.int64)
    change = np.flatnonzero(column[1:] != column[:-1]) + 1
    starts = np.concatenate(([0], change))
    lengths = np.diff(np.concatenate((starts, [n])))
    if lengths.max() > max_run_length:  # split over-long runs
        pieces = div_rounding_up(lengths, max_run_length)
        rep_starts = np.repeat(starts, pieces)
        within = np.arange(pieces.sum()) - np.repeat(np.cumsum(pieces) - pieces, pieces)
        starts = rep_starts + within * max_run_length
        lengths = np.diff(np.concatenate((starts, [n])))
    return starts, lengths


def run_length_encoding(column, segment_length, run_length_dtype=np.uint16, index_size=4):
    column = np.asarray(column)
    n = len(column)
    run_length_dtype = np.dtype(run_length_dtype)
    starts, lengths = _runs_of(column, np.iinfo(run_length_dtype).max)
    seg_starts = np.arange(0, n, segment_length)
    anchor_runs = np.searchsorted(starts, seg_starts, side="right") - 1
    intra = seg_starts - starts[anchor_runs]
    return {
        "run_data": column[starts].copy(),
        "run_lengths": lengths.astype(run_length_dtype),
        "position_anchors": anchor_runs.astype(uint_t(index_size)),
        "intra_run_anchor_offsets": intra.astype(run_length_dtype),
        "segment_length": segment_length,
        "num_segments": len(seg_starts),
        "num_element_runs": len(starts),
        "length": n,
    }


# ------------------------------------------------------------------------------------------------
# Run-position encoding  (src/kernels/decompression/run_position_encoding.cuh:255-308, :311-326)
# ------------------------------------------------------------------------------------------------
def run_position_encoding(column, segment_length, position_dtype=np.uint16, relative=True, index_size=4):
    column = np.asarray(column)
    n = len(column)
    position_dtype = np.dtype(position_dtype)
    starts, _ = _runs_of(column, n + 1)
    if relative:
        if segment_length > np.iinfo(position_dtype).max + 1:
            raise ValueError("segment too long for relative positions of this width")
        positions = starts % segment_length
    else:
        positions = starts
    seg_starts = np.arange(0, n, segment_length)
    anchor_runs = np.searchsorted(starts, seg_starts, side="right") - 1
    return {
        "run_data": column[starts].copy(),
        "run_start_positions": positions.astype(position_dtype),
        "position_anchors": anchor_runs.astype(uint_t(index_size)),
        "segment_length": segment_length,
        "num_anchors": len(seg_starts),
        "num_element_runs": len(starts),
        "length": n,
    }


# ------------------------------------------------------------------------------------------------
# Patches: naive (src/kernels/data_layout/scatter.cuh:13-31) and compressed-indices
# (src/kernels/data_layout/compressed_indices_scatter.cuh:131-217)
# ------------------------------------------------------------------------------------------------
def patches(decoded, truth, index_size=4):
    """Positions where `decoded` differs from `truth`, and the true values there."""
    idx = np.flatnonzero(np.asarray(decoded) != np.asarray(truth))
    return {
        "indices": idx.astype(uint_t(index_size)),
        "data": np.asarray(truth)[idx].copy(),
        "data_length": len(idx),
    }


def compressed_indices(indices, data, anchoring_period, output_index_size=4, input_index_size=4,
                       run_length_size=None, max_run_length=None):
    """Sorted, distinct target positions -> runs of (baseline, offset width in BITS, length, start in
    the byte pool); a new run starts at every multiple of `anchoring_period` patches (:148-160);
    each run's offsets start at an OutputIndexSize-aligned byte (:198-201)."""
    indices = np.asarray(indices, dtype=np.uint64)
    n = len(indices)
    assert n > 0, "zero-length scatters are rejected (compressed_indices_scatter.cuh:288-290)"
    assert np.all(np.diff(indices.astype(np.int64)) > 0), "indices must be sorted and distinct"
    run_length_size = run_length_size or input_index_size
    cap = np.iinfo(uint_t(run_length_size)).max
    if max_run_length:
        cap = min(cap, max_run_length)
    run_lengths, widths, baselines, start_pos, anchors = [], [], [], [], []
    pool = bytearray()
    for a in range(0, n, anchoring_period):
        anchors.append(len(run_lengths))
        block = indices[a:a + anchoring_period]
        i = 0
        while i < len(block):
            base = int(block[i])
            # greedy: grow the run while the width class of (idx - base) stays the same
            def width(v):
                d = int(v) - base
                return 8 if d < (1 << 8) else 16 if d < (1 << 16) else 32 if d < (1 << 32) else 64
            w = width(block[min(i + 1, len(block) - 1)])
            j = i + 1
            while j < len(block) and j - i < cap and width(block[j]) <= w:
                j += 1
            offs = (block[i:j] - np.uint64(base)).astype({8: np.uint8, 16: np.uint16, 32: np.uint32, 64: np.uint64}[w])
            while len(pool) % output_index_size:
                pool.append(0)
            start_pos.append(len(pool) // output_index_size)
            pool.extend(offs.tobytes())
            run_lengths.append(j - i)
            widths.append(w)
            baselines.append(base)
            i = j
    while len(pool) % 8:
        pool.append(0)
    return {
        "data_to_scatter": np.asarray(data).copy(),
        "scatter_position_run_lengths": np.asarray(run_lengths, dtype=uint_t(run_length_size)),
        "scatter_position_run_individual_offset_sizes": np.asarray(widths, dtype=np.uint8),
        "scatter_position_run_baseline_values": np.asarray(baselines, dtype=uint_t(output_index_size)),
        "scatter_position_run_offsets_start_pos": np.asarray(start_pos, dtype=uint_t(input_index_size)),
        "scatter_position_offset_bytes": np.frombuffer(bytes(pool), dtype=np.uint8).copy(),
        "scatter_position_anchors": np.asarray(anchors, dtype=uint_t(input_index_size)),
        "anchoring_period": anchoring_period,
        "num_scatter_position_runs": len(run_lengths),
        "input_data_length": n,
    }
