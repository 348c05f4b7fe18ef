"""Times EVERY registered decompressor instantiation (all widths, both index sizes) on device-generated
synthetic columns and checks each against an independent torch computation, bit for bit.

    python tools/sweep_all.py [log2n=27] [--json out.json]

Prints one line per factory key: ms, algorithmic GB/s, % of the 8 TB/s HBM peak.  A development /
coverage aid: bench.py remains the contract benchmark.  (Scatter kernels are timed on 0.1 % patches.)
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np
import torch

from libgiddy_amd import registered as r, schemes
from workloads import output_buffer

DEV = "cuda"
INT = {1: torch.int8, 2: torch.int16, 4: torch.int32, 8: torch.int64}
BITS = {1: 0xFF, 2: 0xFFFF, 4: 0xFFFFFFFF}


def gen(seed):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    return g


def rand_bits(n, size, g, high=None):
    """n random unsigned values of `size` bytes as the matching signed torch dtype (same bit pattern)"""
    if size == 8:
        return torch.randint(-(1 << 62), 1 << 62, (n,), dtype=torch.int64, generator=g, device=DEV) if high is None else \
            torch.randint(0, high, (n,), dtype=torch.int64, generator=g, device=DEV)
    hi = (1 << (8 * size)) if high is None else high
    v = torch.randint(0, hi, (n,), dtype=torch.int64, generator=g, device=DEV)
    return v.to(INT[size])   # wraps to the signed view


def unsigned(t, size):
    return t.to(torch.int64) & BITS[size] if size < 8 else t


def timeit(fn, reps=int(os.environ.get("SWEEP_REPS", "7"))):
    fn(); fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) for a, b in ev)
    return t[len(t) // 2]


def chunks(n, step=1 << 26):
    for lo in range(0, n, step):
        yield lo, min(n, lo + step)


def make_patches(n, U, I, seed):
    """0.1 % ascending patch positions (I-byte indices) and U-byte values"""
    g = gen(seed)
    count = max(1, int(n * float(os.environ.get("PATCH_FRACTION", "0.001"))))
    idx = torch.sort(torch.randperm(n, generator=g, device=DEV)[:count])[0]
    return idx, idx.to(INT[I]), rand_bits(count, U, g)


def patched_ok(expect, out, idx, vals, lo, hi):
    """expect (the plain decode of [lo, hi)) with the patches inside the range applied == out[lo:hi]"""
    inside = (idx >= lo) & (idx < hi)
    expect = expect.clone()
    expect[idx[inside] - lo] = vals[inside]
    return bool((expect == out[lo:hi]).all())


def run_dictionary(I, U, D, n, patched=False, entries_n=None):
    g = gen(U * 10 + D)
    entries_n = entries_n or min(1 << (8 * D), 4096 if D > 1 else 256)
    entries = rand_bits(entries_n, U, g)
    keys = rand_bits(n, D, g, high=entries_n)
    out = output_buffer(n, INT[U], DEV)
    st = torch.cuda.current_stream().cuda_stream
    if patched:
        idx, pidx, pval = make_patches(n, U, I, U + D)
        kw = dict(uncompressed_size=U, dictionary_index_size=D, index_size=I, stream=st)
        cfg = schemes.patched_dictionary(out, keys, entries, pval, pidx, n, entries_n, len(idx), **kw)
        ms = timeit(lambda: schemes.patched_dictionary(out, keys, entries, pval, pidx, n, entries_n, len(idx), launch_config=cfg, **kw))
        ok = all(patched_ok(entries[unsigned(keys[lo:hi], D)], out, idx, pval, lo, hi) for lo, hi in chunks(n))
        return ms, (U + D) * n + (U + I) * len(idx), ok
    cfg = schemes.dictionary(out, keys, entries, n, entries_n, uncompressed_size=U, dictionary_index_size=D, index_size=I, stream=st)
    ms = timeit(lambda: schemes.dictionary(out, keys, entries, n, entries_n, uncompressed_size=U, dictionary_index_size=D, index_size=I, stream=st, launch_config=cfg))
    ok = all(bool((entries[unsigned(keys[lo:hi], D)] == out[lo:hi]).all()) for lo, hi in chunks(n))
    return ms, (U + D) * n, ok


def run_delta(I, U, C, n, L=4096, patched=False, in_parts=False):
    g = gen(U * 10 + C)
    d = rand_bits(n, C, g)
    du = unsigned(d, C)
    per_seg = du.view(n // L, L).sum(dim=1)
    anchors64 = (1 << 40) + torch.cumsum(per_seg, 0) - per_seg
    anchors = anchors64.to(INT[U])
    out = output_buffer(n, INT[U], DEV)
    st = torch.cuda.current_stream().cuda_stream
    kw = dict(uncompressed_size=U, compressed_size=C, index_size=I, stream=st)
    if in_parts:   # the wrapper's extension for long segments
        kw["part_anchors_scratch"] = torch.zeros(n // 4096 + 1024, dtype=INT[U], device=DEV)
    if patched:
        idx, pidx, pval = make_patches(n, U, I, U + C)
        cfg = schemes.patched_delta(out, d, anchors, pval, pidx, n, L, len(idx), **kw)
        ms = timeit(lambda: schemes.patched_delta(out, d, anchors, pval, pidx, n, L, len(idx), launch_config=cfg, **kw))
    else:
        cfg = schemes.delta(out, d, anchors, n, L, **kw)
        ms = timeit(lambda: schemes.delta(out, d, anchors, n, L, launch_config=cfg, **kw))
    ok = True
    for lo, hi in chunks(n, max(L, (1 << 26) // L * L)):
        expect = (torch.cumsum(du[lo:hi].view(-1, L), dim=1) + anchors64[lo // L: hi // L, None]).view(-1).to(INT[U])
        ok = ok and (patched_ok(expect, out, idx, pval, lo, hi) if patched else bool((expect == out[lo:hi]).all()))
    return ms, (U + C) * n + U * (n // L) + ((U + I) * len(idx) if patched else 0), ok


def run_dzb_fixed(I, U, C, n, patched=False):
    c = rand_bits(n, C, gen(U + C))
    out = output_buffer(n, INT[U], DEV)
    st = torch.cuda.current_stream().cuda_stream
    kw = dict(uncompressed_size=U, compressed_size=C, index_size=I, stream=st)
    if patched:
        idx, pidx, pval = make_patches(n, U, I, U + C + 50)
        cfg = schemes.patched_discard_zero_bytes_fixed(out, c, pval, pidx, n, len(idx), **kw)
        ms = timeit(lambda: schemes.patched_discard_zero_bytes_fixed(out, c, pval, pidx, n, len(idx), launch_config=cfg, **kw))
        ok = all(patched_ok(unsigned(c[lo:hi], C).to(INT[U]), out, idx, pval, lo, hi) for lo, hi in chunks(n))
        return ms, (U + C) * n + (U + I) * len(idx), ok
    cfg = schemes.discard_zero_bytes_fixed(out, c, n, **kw)
    ms = timeit(lambda: schemes.discard_zero_bytes_fixed(out, c, n, launch_config=cfg, **kw))
    ok = bool((unsigned(c, C).to(INT[U]) == out).all())
    return ms, (U + C) * n, ok


def run_dzb_variable(I, U, CS, n, L=1024, bits=2, min_size=1):
    g = gen(U + CS)
    per = (8 * CS) // bits
    hi_size = min(min_size + (1 << bits) - 1, U)
    sizes = torch.randint(min_size, hi_size + 1, (n,), dtype=torch.int64, generator=g, device=DEV)
    offsets = torch.cumsum(sizes, 0) - sizes
    total = int(offsets[-1] + sizes[-1])
    values = torch.randint(0, 1 << 62, (n,), dtype=torch.int64, generator=g, device=DEV)
    mask = torch.where(sizes >= 8, torch.full_like(sizes, -1), (torch.ones_like(sizes) << (8 * sizes)) - 1)
    values = values & mask
    ncont = (n + per - 1) // per
    fields = torch.zeros(ncont * per, dtype=torch.int64, device=DEV)
    fields[:n] = sizes - min_size
    shifts = torch.arange(per, device=DEV, dtype=torch.int64) * bits
    packed64 = (fields.view(ncont, per) << shifts).sum(dim=1)
    packed = packed64.to(INT[CS])
    data = torch.empty(total, dtype=torch.uint8, device=DEV)
    for k in range(hi_size):
        sel = sizes > k
        data[offsets[sel] + k] = ((values[sel] >> (8 * k)) & 0xFF).to(torch.uint8)
    anchors = offsets[torch.arange(0, n, L, device=DEV)].to(INT[I]).contiguous()
    out = output_buffer(n, INT[U], DEV)
    st = torch.cuda.current_stream().cuda_stream
    kw = dict(uncompressed_size=U, container_size=CS, index_size=I, stream=st)
    cfg = schemes.discard_zero_bytes_variable(out, data, packed, anchors, L, n, min_size, bits, **kw)
    ms = timeit(lambda: schemes.discard_zero_bytes_variable(out, data, packed, anchors, L, n, min_size, bits, launch_config=cfg, **kw))
    ok = bool((values.to(INT[U]) == out).all())
    return ms, total + CS * ncont + U * n + I * len(anchors), ok


def run_for(I, U, C, model, MI, n, L=4096, patched=False):
    g = gen(U + C)
    comp = rand_bits(n, C, g)
    dim = 1 if model == "constant" else 2
    coefs = torch.randint(-1000000, 1000000, (n // L, dim), dtype=torch.int32, generator=g, device=DEV)
    if dim == 2:
        coefs[:, 1] = torch.randint(-50, 50, (n // L,), dtype=torch.int32, generator=g, device=DEV)
    out = output_buffer(n, INT[U], DEV)
    st = torch.cuda.current_stream().cuda_stream
    kw = dict(uncompressed_size=U, compressed_size=C, model=model, index_size=I, model_index_size=MI, stream=st)
    if patched:
        idx, pidx, pval = make_patches(n, U, I, U + C)
        cfg = schemes.patched_frame_of_reference(out, comp, coefs, pval, pidx, n, L, len(idx), **kw)
        ms = timeit(lambda: schemes.patched_frame_of_reference(out, comp, coefs, pval, pidx, n, L, len(idx), launch_config=cfg, **kw))
    else:
        cfg = schemes.frame_of_reference(out, comp, coefs, n, L, **kw)
        ms = timeit(lambda: schemes.frame_of_reference(out, comp, coefs, n, L, launch_config=cfg, **kw))
    pos = torch.arange(L, device=DEV, dtype=torch.int64)
    ok = True
    for lo, hi in chunks(n, max(L, (1 << 26) // L * L)):
        c = coefs[lo // L: hi // L].to(torch.int64)
        pred = c[:, :1] + (c[:, 1:2] * pos[None, :] if dim == 2 else 0)
        pred = pred.to(torch.int32).to(INT[U]).to(torch.int64)       # trunc to T = int32, then to Uncompressed
        expect = (pred.expand(-1, L).reshape(-1) + comp[lo:hi].to(torch.int64)).to(INT[U])
        ok = ok and (patched_ok(expect, out, idx, pval, lo, hi) if patched else bool((expect == out[lo:hi]).all()))
    return ms, (U + C) * n + 4 * dim * (n // L) + ((U + I) * len(idx) if patched else 0), ok


def run_model(I, U, model, n, patched=False):
    coefs = np.array([-12345, 37] if model == "linear" else [-4321], dtype="i%d" % U)
    out = output_buffer(n, INT[U], DEV)
    st = torch.cuda.current_stream().cuda_stream
    kw = dict(uncompressed_size=U, model=model, index_size=I, stream=st)
    if patched:
        idx, pidx, pval = make_patches(n, U, I, U + 40)
        cfg = schemes.patched_model(out, coefs, pval, pidx, n, len(idx), **kw)
        ms = timeit(lambda: schemes.patched_model(out, coefs, pval, pidx, n, len(idx), launch_config=cfg, **kw))
    else:
        cfg = schemes.model(out, coefs, n, **kw)
        ms = timeit(lambda: schemes.model(out, coefs, n, launch_config=cfg, **kw))
    ok = True
    for lo, hi in chunks(n):
        p = torch.arange(lo, hi, device=DEV, dtype=torch.int64)
        expect = (p * 37 - 12345 if model == "linear" else torch.full_like(p, -4321)).to(INT[U])
        ok = ok and (patched_ok(expect, out, idx, pval, lo, hi) if patched else bool((expect == out[lo:hi]).all()))
    return ms, U * n + ((U + I) * len(idx) if patched else 0), ok


def run_ib(I, U, A, use_dict, n, nb=16):
    g = gen(U)
    words = n // 32
    keys = torch.randint(0, nb, (n,), dtype=torch.int64, generator=g, device=DEV)
    entries = rand_bits(nb, U, g) if use_dict else torch.arange(nb, device=DEV).to(INT[U])
    aligned = (words + A - 1) // A * A
    bitmaps = torch.zeros((nb, aligned), dtype=torch.int32, device=DEV)
    shifts = torch.arange(32, device=DEV, dtype=torch.int64)
    for b in range(nb):
        word = ((keys == b).view(words, 32).to(torch.int64) << shifts).sum(dim=1)
        bitmaps[b, :words] = torch.where(word >= (1 << 31), word - (1 << 32), word).to(torch.int32)
    out = output_buffer(n, INT[U], DEV)
    st = torch.cuda.current_stream().cuda_stream
    kw = dict(uncompressed_size=U, alignment_in_ints=A, use_dictionary=use_dict, index_size=I, stream=st)
    cfg = schemes.incidence_bitmaps(out, bitmaps.view(-1), entries, words, nb, **kw)
    ms = timeit(lambda: schemes.incidence_bitmaps(out, bitmaps.view(-1), entries, words, nb, launch_config=cfg, **kw))
    return ms, U * n + nb * 4 * words, bool((entries[keys] == out).all())


def _runs(n, mean_run, g, max_run):
    est = int(n / mean_run * 1.2) + 1024
    u = torch.rand(est, generator=g, device=DEV, dtype=torch.float64)
    lengths = torch.clamp(torch.floor(torch.log(u) / np.log(1.0 - 1.0 / mean_run)).to(torch.int64) + 1, 1, max_run)
    ends = torch.cumsum(lengths, 0)
    count = int(torch.searchsorted(ends, torch.tensor([n], device=DEV), right=False)[0]) + 1
    lengths = lengths[:count].clone()
    lengths[-1] -= int(ends[count - 1]) - n
    return lengths, torch.cumsum(lengths, 0) - lengths


def run_rle(I, U, LS, RI, n, L=4096, mean_run=32):
    mean_run = int(os.environ.get("MEAN_RUN", mean_run))   # (development: tools/occupancy_sweep.py at other run lengths)
    g = gen(U * 10 + LS)
    lengths, starts = _runs(n, mean_run, g, min((1 << (8 * LS)) - 1 if LS < 8 else 1 << 40, 65535))
    count = len(lengths)
    values = (torch.arange(count, device=DEV) * 2654435761 % (1 << min(8 * U, 62))).to(INT[U])
    seg = torch.arange(0, n, L, device=DEV)
    anchor = torch.searchsorted(starts, seg, right=True) - 1
    out = output_buffer(n, INT[U], DEV)
    st = torch.cuda.current_stream().cuda_stream
    args = (values, lengths.to(INT[LS]), anchor.to(INT[RI]), (seg - starts[anchor]).to(INT[LS]), L, len(seg), count, n)
    kw = dict(uncompressed_size=U, run_length_size=LS, index_size=I, run_index_size=RI, stream=st)
    cfg = schemes.run_length_encoding(out, *args, **kw)
    ms = timeit(lambda: schemes.run_length_encoding(out, *args, launch_config=cfg, **kw))
    ok = all(bool((values[torch.searchsorted(starts, torch.arange(lo, hi, device=DEV), right=True) - 1] == out[lo:hi]).all()) for lo, hi in chunks(n))
    return ms, U * n + (U + LS) * count + (RI + LS) * len(seg), ok


def run_rpe(I, U, P, relative, RI, n, L=4096, mean_run=32):
    g = gen(U * 10 + P)
    lengths, starts = _runs(n, mean_run, g, 65535)
    count = len(lengths)
    values = (torch.arange(count, device=DEV) * 2654435761 % (1 << min(8 * U, 62))).to(INT[U])
    seg = torch.arange(0, n, L, device=DEV)
    anchor = torch.searchsorted(starts, seg, right=True) - 1
    positions = (starts % L if relative else starts).to(INT[P])
    out = output_buffer(n, INT[U], DEV)
    st = torch.cuda.current_stream().cuda_stream
    args = (values, positions, anchor.to(INT[RI]), L, len(seg), count, n)
    kw = dict(uncompressed_size=U, position_size=P, relative=relative, index_size=I, run_index_size=RI, stream=st)
    cfg = schemes.run_position_encoding(out, *args, **kw)
    ms = timeit(lambda: schemes.run_position_encoding(out, *args, launch_config=cfg, **kw))
    ok = all(bool((values[torch.searchsorted(starts, torch.arange(lo, hi, device=DEV), right=True) - 1] == out[lo:hi]).all()) for lo, hi in chunks(n))
    return ms, U * n + (U + P) * count + RI * len(seg), ok


def run_scatter(O, E, IS, n):
    g = gen(O + E)
    p = min(max(1, n // 1000), (1 << (8 * IS)) - 1 if IS < 4 else n)
    idx = torch.sort(torch.randperm(n, generator=g, device=DEV)[:p])[0]
    val = rand_bits(p, E, g)
    target = torch.zeros(n, dtype=INT[E], device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    kw = dict(output_index_size=O, element_size=E, input_index_size=IS, stream=st)
    cfg = schemes.scatter(target, val, idx.to(INT[O]), p, **kw)
    ms = timeit(lambda: schemes.scatter(target, val, idx.to(INT[O]), p, launch_config=cfg, **kw))
    expect = torch.zeros(n, dtype=INT[E], device=DEV)
    expect[idx] = val
    return ms, p * (O + 2 * E), bool((expect == target).all())


_compressed = {}


def _compressed_positions(O, IS, RLS, n):
    """ascending patch positions (0.1 % of the column; as many as the input index type can count) and their compressed-indices form
    (libgiddy_amd.formats, a host loop: cached per index-size combination), on the device"""
    key = (O, IS, RLS, n)
    if key not in _compressed:
        from libgiddy_amd import formats
        g = gen(O + IS)
        p = min(max(1, n // 1000), (1 << (8 * min(IS, RLS))) - 1 if min(IS, RLS) < 4 else n)
        idx = torch.sort(torch.randperm(min(n, (1 << (8 * O)) - 1 if O < 8 else n), generator=g, device=DEV)[:p])[0]
        c = formats.compressed_indices(idx.cpu().numpy().astype(np.uint64), np.zeros(0, np.uint8), 1024 if p > 4096 else 64,
                                       output_index_size=O, input_index_size=IS, run_length_size=RLS)
        on_device = {k: (torch.from_numpy(v.view({1: np.uint8, 2: np.int16, 4: np.int32, 8: np.int64}[v.dtype.itemsize])).to(DEV) if isinstance(v, np.ndarray) and v.size else v)
                     for k, v in c.items() if k != "data_to_scatter"}
        nbytes = sum(v.numel() * v.element_size() for v in on_device.values() if isinstance(v, torch.Tensor))
        _compressed[key] = (idx, on_device, nbytes)
    return _compressed[key]


def run_cis(O, E, IS, RLS, n):
    idx, c, position_bytes = _compressed_positions(O, IS, RLS, n)
    p = len(idx)
    val = rand_bits(p, E, gen(E))
    target = torch.zeros(n, dtype=INT[E], device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    args = (target, val, c["scatter_position_run_lengths"], c["scatter_position_run_individual_offset_sizes"], c["scatter_position_run_baseline_values"],
            c["scatter_position_run_offsets_start_pos"], c["scatter_position_offset_bytes"], c["scatter_position_anchors"], c["anchoring_period"],
            c["num_scatter_position_runs"], p)
    kw = dict(output_index_size=O, element_size=E, input_index_size=IS, run_length_size=RLS, stream=st)
    cfg = schemes.compressed_indices_scatter(*args, **kw)
    ms = timeit(lambda: schemes.compressed_indices_scatter(*args, launch_config=cfg, **kw))
    expect = torch.zeros(n, dtype=INT[E], device=DEV)
    expect[idx] = val
    return ms, position_bytes + 2 * p * E, bool((expect == target).all())


def run_cis_positions(O, IS, RLS, n):
    idx, c, position_bytes = _compressed_positions(O, IS, RLS, n)
    p = len(idx)
    out = torch.zeros(p, dtype=INT[O], device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    kw = dict(output_index_size=O, input_index_size=IS, run_length_size=RLS, stream=st)
    cfg = schemes.compressed_indices_positions(out, c, p, **kw)
    ms = timeit(lambda: schemes.compressed_indices_positions(out, c, p, launch_config=cfg, **kw))
    return ms, position_bytes + p * O, bool((unsigned(out, O) == idx).all()) if O < 8 else bool((out == idx).all())


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 27
    n = 1 << log2n
    results = {}
    num = lambda s: [int(x) for x in re.findall(r"(\d+)u", s)]
    ctype = {"signed char": 1, "short": 2, "int": 4, "long": 8}
    for key in r.kernel_t.registered_keys():
        if key.startswith("compression::"):
            continue   # (the compressors: tools/compress_bench.py)
        if os.environ.get("ONLY") and os.environ["ONLY"] not in key:
            continue   # (development: one family)
        torch.cuda.empty_cache()
        try:
            patched = "::patched::" in key   # the fused decompress-and-patch extensions: 0.1 % patches
            if key.startswith("decompression::dictionary"):
                I, U, D = num(key); res = run_dictionary(I, U, D, n, patched=patched)
            elif key.startswith("decompression::delta"):
                I, U, C = num(key); res = run_delta(I, U, C, n, patched=patched)
            elif "fixed_width" in key:
                I, U, C = num(key)[:3]; res = run_dzb_fixed(I, U, C, n, patched=patched)
            elif "variable_width" in key:
                I, U, CS = num(key); res = run_dzb_variable(I, U, CS, min(n, 1 << 28))
            elif key.startswith("decompression::frame_of_reference"):
                m = re.match(r".*kernel_t<(\d)u, ([a-z ]+), ([a-z ]+), .*::(constant|linear)<(\d)u", key)
                res = run_for(int(m.group(1)), ctype[m.group(2)], ctype[m.group(3)], m.group(4), int(m.group(5)), n, patched=patched)
            elif key.startswith("decompression::model"):
                m = re.match(r".*kernel_t<(\d)u, ([a-z ]+), .*::(constant|linear)<", key)
                res = run_model(int(m.group(1)), ctype[m.group(2)], m.group(3), n, patched=patched)
            elif key.startswith("decompression::incidence_bitmaps"):
                I, U, A = num(key); res = run_ib(I, U, A, key.endswith("true>"), min(n, 1 << 28))
            elif key.startswith("decompression::run_length_encoding"):
                I, U, LS, RI = num(key); res = run_rle(I, U, LS, RI, n)
            elif key.startswith("decompression::run_position_encoding"):
                I, U, P, RI = num(key); res = run_rpe(I, U, P, "true" in key, RI, n)
            elif key.startswith("scatter::kernel_t"):
                O, E, IS = num(key)[:3]; res = run_scatter(O, E, IS, n)
            elif key.startswith("scatter::compressed_indices::positions"):
                O, IS, RLS = num(key); res = run_cis_positions(O, IS, RLS, n)
            elif key.startswith("scatter::compressed_indices::kernel_t"):
                O, E, IS, RLS = num(key); res = run_cis(O, E, IS, RLS, n)
            else:
                continue
        except torch.OutOfMemoryError:
            print("%-110s skipped (memory)" % key); continue
        ms, nbytes, ok = res
        gbps = nbytes / ms / 1e6
        results[key] = {"ms": ms, "algorithmic_GBps": gbps, "pct_peak": gbps / 80.0, "bit_exact": ok}
        print("%-110s %8.3f ms %8.1f GB/s %5.1f%%  %s" % (key, ms, gbps, gbps / 80.0, "ok" if ok else "MISMATCH"), flush=True)
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(results, f, indent=1)
    bad = [k for k, v in results.items() if not v["bit_exact"]]
    print("instantiations swept: %d, mismatches: %d" % (len(results), len(bad)))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
