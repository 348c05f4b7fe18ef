"""Segment-range sharding of compressed columns across the GPUs of a node.

Every variable-rate scheme carries per-segment anchors so that any anchored segment decodes
independently (reference: delta.cuh:5-19, discard_zero_bytes_variable.cuh:355-370,
run_length_encoding.cuh:285-288, run_position_encoding.cuh:255-308,
compressed_indices_scatter.cuh:202-212) and fixed-rate schemes are elementwise — so rank g of W
decodes output elements [start_g, end_g), a whole number of segments, from:
  * the same slice of the fixed-rate inputs,
  * the anchor slice, with anchors REBASED to the shard (run indices / byte offsets start at 0),
  * for RLE/RPE the run slice [anchor[s0], anchor[s1]] (the run straddling the shard's end included),
  * for DZB-variable the byte slice [anchor[s0], anchor[s1]) and the size-container slice — shard
    boundaries are kept multiples of lcm(segment_length, sizes-per-container) so that a shard's
    first size field is field 0 of a container,
  * a replica of small side tables (dictionary, model record).
There is NO exchange step during decode; the only collective is the optional gather of the decoded
shards into one contiguous buffer (libgiddy_amd.distributed.gather_decoded).

Everything here is index arithmetic; the arrays may be numpy arrays, host (pinned) torch tensors or DEVICE
torch tensors, in any mix: the big arrays of a shard are slices — views, nothing is copied — and the only
new arrays are the rebased anchor slices (one small subtraction, done where the anchors live: a column that
is resident on a GPU is cut without a host round trip; reading the two boundary anchors costs two scalar
device-to-host copies).  Integer tensors stand for the unsigned bits they hold (torch has little unsigned
arithmetic): subtraction wraps the same way.  Returns `Shard(fields, start, stop)`; `fields` has the keys of
libgiddy_amd.formats' writers.
"""
import math
from collections import namedtuple

import numpy as np


def _is_tensor(a):
    return type(a).__module__.partition(".")[0] == "torch"


def _itemsize(a):
    return a.element_size() if _is_tensor(a) else a.dtype.itemsize


def _unsigned_at(a, i):
    """a[i] as a Python int, the element's bits read as unsigned"""
    if _is_tensor(a):
        return int(a[i].item()) & ((1 << (8 * a.element_size())) - 1)
    return int(a[i]) & ((1 << (8 * a.dtype.itemsize)) - 1)


def _rebased(a, offset):
    """a - offset in wrapping arithmetic of a's width (numpy: any integer dtype; torch: signed or unsigned integer tensors)"""
    bits = 8 * _itemsize(a)
    offset = int(offset) & ((1 << bits) - 1)
    if _is_tensor(a):
        import torch
        signed = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}[a.element_size()]
        wrapped = offset - (1 << bits) if offset >= (1 << (bits - 1)) else offset
        return (a.view(signed) - wrapped).view(a.dtype)
    return a - a.dtype.type(offset if a.dtype.kind == "u" else (offset - (1 << bits) if offset >= (1 << (bits - 1)) else offset))

Shard = namedtuple("Shard", ["fields", "start", "stop"])   # decodes output elements [start, stop)


def segment_range(num_segments, rank, world_size, granule=1):
    """Contiguous, balanced split of `num_segments` segments into `world_size` ranges whose boundaries
    are multiples of `granule` segments (the last range takes the remainder)."""
    units = -(-num_segments // granule)
    lo = (units * rank // world_size) * granule
    hi = (units * (rank + 1) // world_size) * granule
    return min(lo, num_segments), min(hi, num_segments)


PAGE_BYTES = 64 * 1024   # shard boundaries fall on multiples of this many OUTPUT bytes where segments tile it (see page_granule)


def page_granule(segment_length, output_itemsize, granule=1, page_bytes=None):
    """Segments per shard-boundary granule so that a shard's output begins a whole number of 64 KiB pages into the column's: the
    wave-per-segment decoders write a workgroup's four segments into one such page when the output is placed on a page boundary, and a
    shard that begins 16 or 32 KiB into a page costs them 6 points ("where the output lies"; the kernels now also
    regroup by address, this keeps shards from needing it).  Only where segments tile a page; `granule`: a granule the scheme needs
    anyway (the result is a multiple of it)."""
    page = PAGE_BYTES if page_bytes is None else page_bytes
    segment_bytes = int(segment_length) * int(output_itemsize)
    if segment_bytes <= 0 or page % segment_bytes != 0:
        return granule
    per_page = page // segment_bytes
    return granule * per_page // math.gcd(granule, per_page)


def _element_range(length, segment_length, rank, world_size, granule=1):
    num_segments = -(-length // segment_length)
    s0, s1 = segment_range(num_segments, rank, world_size, granule)
    return s0, s1, min(s0 * segment_length, length), min(s1 * segment_length, length)


def elementwise(fields, rank, world_size, array_keys=("compressed_input",), granule=PAGE_BYTES):
    """Dictionary / DZB-fixed: split on multiples of `granule` elements (64 Ki: a whole number of 64 KiB output pages whatever the
    element size); side tables are replicated."""
    n = int(fields["length"])
    s0, s1, lo, hi = _element_range(n, granule, rank, world_size)
    out = dict(fields)
    for k in array_keys:
        out[k] = fields[k][lo:hi]
    out["length"] = hi - lo
    return Shard(out, lo, hi)


def frame_of_reference(fields, rank, world_size, output_itemsize=4):
    n, L = int(fields["length"]), int(fields["segment_length"])
    s0, s1, lo, hi = _element_range(n, L, rank, world_size, page_granule(L, output_itemsize))
    out = dict(fields)
    out["compressed_input"] = fields["compressed_input"][lo:hi]
    out["segment_model_coefficients"] = fields["segment_model_coefficients"][s0:s1]
    out["length"] = hi - lo
    return Shard(out, lo, hi)


def model(fields, rank, world_size, model="linear", granule=4096):
    """out[p] = model(p): a shard starting at element `lo` uses c0' = c0 + c1*lo (wrapping), c1' = c1."""
    n = int(fields["length"])
    s0, s1, lo, hi = _element_range(n, granule, rank, world_size)
    c = np.array(fields["model_coefficients"]).copy()
    if model == "linear":
        bits = 8 * c.dtype.itemsize
        c0 = (int(c[0]) + int(c[1]) * lo) % (1 << bits)
        c[0] = np.array(c0, dtype=np.dtype("u%d" % c.dtype.itemsize)).view(c.dtype) if c.dtype.kind == "i" else c0
    out = dict(fields, model_coefficients=c, length=hi - lo)
    return Shard(out, lo, hi)


def delta(fields, rank, world_size):
    n, L = int(fields["length"]), int(fields["segment_length"])
    s0, s1, lo, hi = _element_range(n, L, rank, world_size, page_granule(L, _itemsize(fields["anchor_values"])))
    out = dict(fields)
    out["compressed_input"] = fields["compressed_input"][lo:hi]
    out["anchor_values"] = fields["anchor_values"][s0:s1]
    out["length"] = hi - lo
    return Shard(out, lo, hi)


def discard_zero_bytes_variable(fields, rank, world_size, output_itemsize=8):
    n, L = int(fields["length"]), int(fields["segment_length"])
    bits = int(fields["bits_per_element_size"])
    per = (8 * _itemsize(fields["packed_element_sizes"])) // bits
    granule = (L * per // math.gcd(L, per)) // L          # segments per lcm(L, per) elements
    granule = page_granule(L, output_itemsize, granule)
    s0, s1, lo, hi = _element_range(n, L, rank, world_size, granule)
    anchors = fields["position_anchors"]
    total_bytes = len(fields["compressed_data"])
    b0 = _unsigned_at(anchors, s0) if s0 < len(anchors) else total_bytes
    b1 = _unsigned_at(anchors, s1) if s1 < len(anchors) else total_bytes
    out = dict(fields)
    out["compressed_data"] = fields["compressed_data"][b0:b1]
    out["packed_element_sizes"] = fields["packed_element_sizes"][lo // per: -(-hi // per)]
    out["position_anchors"] = _rebased(anchors[s0:s1], b0) if s1 > s0 else anchors[s0:s1]
    out["length"] = hi - lo
    return Shard(out, lo, hi)


def incidence_bitmaps(fields, rank, world_size, alignment_in_ints=2, granule_words=128):
    """Split on whole bitmap words; every bitmap is sliced and re-laid out at the shard's aligned stride."""
    words, nb = int(fields["bitmap_length"]), int(fields["num_bitmaps"])
    aligned = -(-words // alignment_in_ints) * alignment_in_ints
    s0, s1, w0, w1 = _element_range(words, granule_words, rank, world_size)
    shard_aligned = max(-(-(w1 - w0) // alignment_in_ints) * alignment_in_ints, 0)
    bitmaps = fields["incidence_bitmaps"]
    if _is_tensor(bitmaps):
        import torch
        laid_out = torch.zeros((nb, shard_aligned), dtype=bitmaps.dtype, device=bitmaps.device)
        laid_out[:, : w1 - w0] = bitmaps.reshape(nb, aligned)[:, w0:w1]
    else:
        laid_out = np.zeros((nb, shard_aligned), dtype=bitmaps.dtype)
        laid_out[:, : w1 - w0] = np.asarray(bitmaps).reshape(nb, aligned)[:, w0:w1]
    out = dict(fields, incidence_bitmaps=laid_out.reshape(-1), bitmap_length=w1 - w0)
    return Shard(out, 32 * w0, 32 * w1)


def run_length_encoding(fields, rank, world_size):
    n, L = int(fields["length"]), int(fields["segment_length"])
    s0, s1, lo, hi = _element_range(n, L, rank, world_size, page_granule(L, _itemsize(fields["run_data"])))
    anchors = fields["position_anchors"]
    num_runs = int(fields["num_element_runs"])
    if s1 <= s0:
        empty = {k: (v[:0] if hasattr(v, "__len__") else v) for k, v in fields.items()}
        return Shard(dict(empty, length=0, num_segments=0, num_element_runs=0), lo, hi)
    r0 = _unsigned_at(anchors, s0)
    r1 = _unsigned_at(anchors, s1) + 1 if s1 < len(anchors) else num_runs     # the run straddling the end is included
    out = dict(fields)
    out["run_data"] = fields["run_data"][r0:r1]
    out["run_lengths"] = fields["run_lengths"][r0:r1]
    out["position_anchors"] = _rebased(anchors[s0:s1], r0)
    out["intra_run_anchor_offsets"] = fields["intra_run_anchor_offsets"][s0:s1]
    out["num_segments"] = s1 - s0
    out["num_element_runs"] = r1 - r0
    out["length"] = hi - lo
    return Shard(out, lo, hi)


def run_position_encoding(fields, rank, world_size, relative=True):
    n, L = int(fields["length"]), int(fields["segment_length"])
    s0, s1, lo, hi = _element_range(n, L, rank, world_size, page_granule(L, _itemsize(fields["run_data"])))
    anchors = fields["position_anchors"]
    num_runs = int(fields["num_element_runs"])
    if s1 <= s0:
        empty = {k: (v[:0] if hasattr(v, "__len__") else v) for k, v in fields.items()}
        return Shard(dict(empty, length=0, num_anchors=0, num_element_runs=0), lo, hi)
    r0 = _unsigned_at(anchors, s0)
    r1 = _unsigned_at(anchors, s1) + 1 if s1 < len(anchors) else num_runs
    positions = fields["run_start_positions"][r0:r1]
    if not relative:
        # absolute positions are rebased to the shard; the first run may have started before the shard —
        # its (wrapped) position is never read (run_position_encoding.cuh:122,221)
        positions = _rebased(positions, lo)
    out = dict(fields)
    out["run_data"] = fields["run_data"][r0:r1]
    out["run_start_positions"] = positions
    out["position_anchors"] = _rebased(anchors[s0:s1], r0)
    out["num_anchors"] = s1 - s0
    out["num_element_runs"] = r1 - r0
    out["length"] = hi - lo
    return Shard(out, lo, hi)


def patches(fields, start, stop):
    """Naive patches whose target lies in the shard [start, stop), positions rebased to the shard.  ASCENDING positions
    (what the fused decoders need anyway) are cut with two binary searches — views again; any other order is filtered."""
    idx, data = fields["indices"], fields["data"]
    n = len(idx)
    if _is_tensor(idx):
        import torch
        wide = idx.to(torch.int64) & ((1 << (8 * idx.element_size())) - 1) if idx.element_size() < 8 else idx.view(torch.int64)
        if n < 2 or bool((wide[1:] >= wide[:-1]).all()):
            lo, hi = (int(x) for x in torch.searchsorted(wide, torch.tensor([start, stop], dtype=torch.int64, device=wide.device)))
            return {"indices": _rebased(idx[lo:hi], start), "data": data[lo:hi], "data_length": hi - lo}
        inside = (wide >= start) & (wide < stop)
        return {"indices": _rebased(idx[inside], start), "data": data[inside], "data_length": int(inside.sum())}
    idx = np.asarray(idx)
    if n < 2 or bool((idx[1:] >= idx[:-1]).all()):
        lo, hi = (int(x) for x in np.searchsorted(idx, [start, stop]))
        return {"indices": _rebased(idx[lo:hi], start), "data": np.asarray(data)[lo:hi], "data_length": hi - lo}
    inside = (idx >= start) & (idx < stop)
    return {"indices": _rebased(idx[inside], start), "data": np.asarray(data)[inside], "data_length": int(inside.sum())}
