"""Second, deliberately naive statement of every scheme in numpy (np.cumsum / np.repeat / fancy
indexing / byte slicing), written independently of giddy_oracle.c.  TEST INFRASTRUCTURE ONLY.

It is what tests/golden/make_golden.py runs to produce the committed fixtures, and what the CPU
tests cross-check the C oracle against.  Semantics follow SURVEY.md Appendix A, which cites the
reference file:line for each definition.
"""
import numpy as np


def _wrap(values, dtype):
    """Two's-complement truncation of python/np.int64/object integers to dtype."""
    dtype = np.dtype(dtype)
    bits = 8 * dtype.itemsize
    v = np.asarray(values, dtype=object) % (1 << bits)
    out = np.array([int(x) for x in v.ravel()], dtype=np.uint64).astype(np.dtype("u%d" % dtype.itemsize))
    return out.view(dtype).reshape(np.shape(values))


def frame_of_reference(compressed, coefficients, length, segment_length, out_dtype, model="constant",
                       model_index_size=4):
    # frame_of_reference.cuh:49-79, model_functors.hpp:87-90,108-111
    p = np.arange(length)
    s = p // segment_length
    x = (p % segment_length) % (1 << (8 * model_index_size))
    coefficients = np.asarray(coefficients).reshape(-1, 1 if model == "constant" else 2)
    cdt = coefficients.dtype
    c = coefficients.astype(object)
    t = c[s, 0] if model == "constant" else c[s, 0] + c[s, 1] * x.astype(object)
    t = _wrap(t, cdt).astype(object)          # return type T of the model
    t = _wrap(t, out_dtype).astype(object)    # static_cast<Uncompressed>
    return _wrap(t + np.asarray(compressed[:length]).astype(object), out_dtype)


def model(coefficients, length, out_dtype, model="linear", model_index_size=4):
    # model.cuh:44-52
    coefficients = np.asarray(coefficients).reshape(-1)
    x = (np.arange(length) % (1 << (8 * model_index_size))).astype(object)
    c = coefficients.astype(object)
    t = np.full(length, c[0], dtype=object) if model == "constant" else c[0] + c[1] * x
    return _wrap(_wrap(t, coefficients.dtype).astype(object), out_dtype)


def dictionary(keys, dictionary_entries):
    # gather.cuh:138-144
    return dictionary_entries[keys.astype(np.int64)]


def delta(compressed, anchors, segment_length):
    # delta.cuh:79-107 — inclusive scan of unsigned deltas per segment, seeded with the anchor
    n = len(compressed)
    out = np.zeros(n, dtype=anchors.dtype)
    for s in range(len(anchors)):
        lo, hi = s * segment_length, min((s + 1) * segment_length, n)
        if lo >= n:
            break
        out[lo:hi] = anchors[s] + np.cumsum(compressed[lo:hi].astype(anchors.dtype), dtype=anchors.dtype)
    return out


def dzb_fixed(compressed, out_dtype):
    # discard_zero_bytes_fixed.cuh:54-76
    return compressed.astype(out_dtype)


def dzb_variable(compressed_data, packed_sizes, anchors, segment_length, length, min_size, bits, out_dtype):
    # discard_zero_bytes_variable.cuh:44-58,196-213,397-438
    out_dtype = np.dtype(out_dtype)
    per = (8 * packed_sizes.dtype.itemsize) // bits
    i = np.arange(length)
    fields = (packed_sizes[i // per].astype(np.uint64) >> ((i % per) * bits).astype(np.uint64)) & np.uint64((1 << bits) - 1)
    sizes = fields.astype(np.int64) + min_size
    out = np.zeros(length, dtype=out_dtype)
    for s in range(len(anchors)):
        lo, hi = s * segment_length, min((s + 1) * segment_length, length)
        if lo >= length:
            break
        off = int(anchors[s])
        for e in range(lo, hi):
            n = int(sizes[e])
            out[e] = int.from_bytes(bytes(compressed_data[off:off + n]), "little")
            off += n
    return out


def incidence_bitmaps(bitmaps, dictionary_entries, bitmap_length, num_bitmaps, alignment, out_dtype,
                      use_dictionary=True, fill=0):
    # incidence_bitmaps.cuh:47-60,94-126 — ascending bitmap index, so the highest index wins
    aligned = -(-bitmap_length // alignment) * alignment
    bm = np.asarray(bitmaps).reshape(num_bitmaps, aligned)[:, :bitmap_length]
    out = np.full(32 * bitmap_length, fill, dtype=out_dtype)
    for b in range(num_bitmaps):
        bits = ((bm[b][:, None] >> np.arange(32, dtype=np.uint32)[None, :]) & 1).astype(bool).reshape(-1)
        out[bits] = dictionary_entries[b] if use_dictionary else b
    return out


def run_length_encoding(run_data, run_lengths, length):
    # run_length_encoding.cuh:129-157 — plain expansion (anchors are redundant information)
    return np.repeat(run_data, run_lengths.astype(np.int64))[:length]


def run_position_encoding(run_data, run_start_positions, anchors, segment_length, length, relative):
    # run_position_encoding.cuh:111-150 — absolute starts are recovered, then plain expansion
    nruns = len(run_data)
    pos = run_start_positions.astype(np.int64)
    if relative:
        # the segment a run starts in: runs (anchors[s], anchors[s+1]] start in s, except that a
        # run with relative position 0 starts its own segment (and is that segment's anchor run)
        seg_of_run = np.zeros(nruns, dtype=np.int64)
        a = anchors.astype(np.int64)
        for s in range(len(a)):
            hi = a[s + 1] if s + 1 < len(a) else nruns - 1
            seg_of_run[a[s] + 1:hi + 1] = s
        zero = np.flatnonzero(pos == 0)
        # a run at relative position 0 starts exactly at a segment boundary: find which one
        for r in zero:
            cands = np.flatnonzero(a == r)
            seg_of_run[r] = cands[0] if len(cands) else seg_of_run[r]
        starts = seg_of_run * segment_length + pos
    else:
        starts = pos
    lengths = np.diff(np.concatenate((starts, [length])))
    return np.repeat(run_data, lengths)[:length]


def scatter(target, data, indices):
    # scatter.cuh:42-45
    target = target.copy()
    target[indices.astype(np.int64)] = data
    return target


def compressed_indices_scatter(target, data, run_lengths, offset_sizes, baselines, start_pos, offset_bytes,
                               anchors, anchoring_period, output_index_size):
    # compressed_indices_scatter.cuh:83-124,236-269
    target = target.copy()
    n = len(data)
    nblocks = -(-n // anchoring_period)
    for a in range(nblocks):
        first = int(anchors[a])
        last = len(run_lengths) if (a + 1) * anchoring_period >= n else int(anchors[a + 1])
        k = a * anchoring_period
        for r in range(first, last):
            w = int(offset_sizes[r])
            ln = int(run_lengths[r])
            if w in (8, 16, 32, 64):
                begin = int(start_pos[r]) * output_index_size
                offs = np.frombuffer(bytes(offset_bytes[begin:begin + ln * w // 8]), dtype="<u%d" % (w // 8))
                target[(int(baselines[r]) + offs.astype(np.int64))] = data[k:k + ln]
            k += ln
    return target


def compressed_indices_positions(num_patches, run_lengths, offset_sizes, baselines, start_pos, offset_bytes,
                                 anchors, anchoring_period, output_index_size):
    """positions[k] = where compressed_indices_scatter() puts data[k] (all-ones where it skips a run): the position half of
    compressed_indices_scatter.cuh:83-124,236-269 — what the fused decoders are handed as plain ascending positions"""
    positions = np.full(num_patches, np.iinfo("u%d" % output_index_size).max, dtype="u%d" % output_index_size)
    nblocks = -(-num_patches // anchoring_period)
    for a in range(nblocks):
        first = int(anchors[a])
        last = len(run_lengths) if (a + 1) * anchoring_period >= num_patches else int(anchors[a + 1])
        k = a * anchoring_period
        for r in range(first, last):
            w = int(offset_sizes[r])
            ln = int(run_lengths[r])
            if w in (8, 16, 32, 64):
                begin = int(start_pos[r]) * output_index_size
                offs = np.frombuffer(bytes(offset_bytes[begin:begin + ln * w // 8]), dtype="<u%d" % (w // 8))
                positions[k:k + ln] = int(baselines[r]) + offs.astype(np.uint64)
            k += ln
    return positions
