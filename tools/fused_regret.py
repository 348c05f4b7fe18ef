"""GPU box: the fused decompress-and-patch wrappers against what they replace — the plain decoder followed by data_layout::scatter, two launches on one stream —
over patch densities, for the instantiations DESIGN.md lists as fused patching's weak spots and for each family's usual one.

    python tools/fused_regret.py [log2 n = 28] [--densities 0.001,0.01,...] [--only PATTERN]

Per shape: % of the 8 TB/s HBM peak over the FUSED algorithmic bytes ((U + C) n + (U + I) patches) for both ways, so that the two columns compare; `regret` =
two launches - fused, in points, where the two launches are the faster.  The fused result is checked against decode + scatter bit for bit."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

import numpy as np
import torch

import sweep_all as sw
from libgiddy_amd import schemes
from workloads import output_buffer

INT, DEV = sw.INT, sw.DEV


def patches(n, U, I, density, seed):
    g = sw.gen(seed)
    count = max(1, int(n * density))
    # (uniformly random ascending positions; a draw of n * density positions with replacement, made distinct: randperm of 2^28 is 2 GiB per call)
    idx = torch.unique(torch.randint(0, n, (count,), generator=g, device=DEV, dtype=torch.int64))
    return idx.to(INT[I]), sw.rand_bits(len(idx), U, g)


def shapes(n):
    """(name, U, I, fused algorithmic bytes without patches, fused(out, pval, pidx, count), plain(out))"""
    st = torch.cuda.current_stream().cuda_stream
    L = 4096

    def dictionary(I, U, D, entries_n):
        g = sw.gen(U * 10 + D)
        entries = sw.rand_bits(entries_n, U, g); keys = sw.rand_bits(n, D, g, high=entries_n)
        kw = dict(uncompressed_size=U, dictionary_index_size=D, index_size=I, stream=st)
        return ("dictionary<%d,%d,%d> %d entries" % (I, U, D, entries_n), U, I, (U + D) * n,
                lambda out, pv, pi, c: schemes.patched_dictionary(out, keys, entries, pv, pi, n, entries_n, c, **kw),
                lambda out: schemes.dictionary(out, keys, entries, n, entries_n, **kw))

    def delta(I, U, C):
        g = sw.gen(U * 10 + C)
        d = sw.rand_bits(n, C, g)
        anchors = sw.rand_bits(n // L, U, g)
        kw = dict(uncompressed_size=U, compressed_size=C, index_size=I, stream=st)
        return ("delta<%d,%d,%d>" % (I, U, C), U, I, (U + C) * n + U * (n // L),
                lambda out, pv, pi, c: schemes.patched_delta(out, d, anchors, pv, pi, n, L, c, **kw),
                lambda out: schemes.delta(out, d, anchors, n, L, **kw))

    def dzb_fixed(I, U, C):
        c_in = sw.rand_bits(n, C, sw.gen(U + C))
        kw = dict(uncompressed_size=U, compressed_size=C, index_size=I, stream=st)
        return ("discard_zero_bytes_fixed<%d,%d,%d>" % (I, U, C), U, I, (U + C) * n,
                lambda out, pv, pi, c: schemes.patched_discard_zero_bytes_fixed(out, c_in, pv, pi, n, c, **kw),
                lambda out: schemes.discard_zero_bytes_fixed(out, c_in, n, **kw))

    def frame_of_reference(I, U, C, model):
        g = sw.gen(U + C)
        comp = sw.rand_bits(n, C, g)
        dim = 1 if model == "constant" else 2
        coefs = torch.randint(-1000000, 1000000, (n // L, dim), dtype=torch.int32, generator=g, device=DEV)
        kw = dict(uncompressed_size=U, compressed_size=C, model=model, index_size=I, model_index_size=4, stream=st)
        return ("frame_of_reference<%d,%d,%d,%s>" % (I, U, C, model), U, I, (U + C) * n + 4 * dim * (n // L),
                lambda out, pv, pi, c: schemes.patched_frame_of_reference(out, comp, coefs, pv, pi, n, L, c, **kw),
                lambda out: schemes.frame_of_reference(out, comp, coefs, n, L, **kw))

    def model(I, U, which):
        coefs = np.array([-12345, 37] if which == "linear" else [-4321], dtype="i%d" % U)
        kw = dict(uncompressed_size=U, model=which, index_size=I, stream=st)
        return ("model<%d,%d,%s>" % (I, U, which), U, I, U * n,
                lambda out, pv, pi, c: schemes.patched_model(out, coefs, pv, pi, n, c, **kw),
                lambda out: schemes.model(out, coefs, n, **kw))

    yield lambda: dictionary(4, 4, 1, 256)
    yield lambda: dictionary(4, 8, 2, 4096)      # a 32 KiB dictionary staged beside the patch windows (round 5: 28 % at 2 % of patches)
    yield lambda: dictionary(4, 8, 2, 2048)      # 16 KiB staged: where the rule begins
    yield lambda: dictionary(4, 8, 2, 1024)
    yield lambda: dictionary(4, 8, 4, 4096)
    yield lambda: dictionary(4, 4, 2, 8192)      # 32 KiB of 4-byte values
    yield lambda: dictionary(4, 8, 1, 256)
    yield lambda: dictionary(4, 2, 1, 256)
    yield lambda: delta(4, 8, 2)
    yield lambda: delta(4, 4, 1)
    yield lambda: delta(4, 2, 1)                 # 2-byte outputs: VALU-bound rows carrying twice the patches (round 5: 41-55 % at 0.25-2 %)
    yield lambda: frame_of_reference(4, 4, 2, "linear")
    yield lambda: frame_of_reference(4, 2, 1, "constant")
    yield lambda: dzb_fixed(4, 4, 1)
    yield lambda: dzb_fixed(4, 2, 1)
    yield lambda: dzb_fixed(4, 8, 2)
    yield lambda: model(4, 4, "linear")
    yield lambda: model(4, 2, "linear")


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else 28
    n = 1 << log2n
    densities = [float(x) for x in (sys.argv[sys.argv.index("--densities") + 1].split(",") if "--densities" in sys.argv else
                                    "0.001 0.0025 0.005 0.01 0.02 0.05 0.1 0.2 0.5".split())]
    only = re.compile(sys.argv[sys.argv.index("--only") + 1]) if "--only" in sys.argv else None
    st = torch.cuda.current_stream().cuda_stream
    print("# tools/fused_regret.py 2^%d elements: %% of the HBM peak (fused algorithmic bytes) — fused wrapper | plain decoder + scatter (two launches); regret = points the wrapper loses" % log2n)
    worst = (0.0, None)
    for make in shapes(n):
        name, U, I, base_bytes, fused, plain = make()
        if only and not only.search(name):
            continue
        out, check = output_buffer(n, INT[U], DEV), output_buffer(n, INT[U], DEV)
        for density in densities:
            pidx, pval = patches(n, U, I, density, int(density * 1e6) + U)
            count = len(pidx)
            skw = dict(output_index_size=I, element_size=U, input_index_size=I, stream=st)
            nbytes = base_bytes + (U + I) * count
            fused_config = fused(out, pval, pidx, count)
            plain_config = plain(check)
            scatter_config = schemes.scatter(check, pval, pidx, count, **skw)
            torch.cuda.synchronize()
            ok = bool((out == check).all())

            def two_launches():
                plain(check)
                schemes.scatter(check, pval, pidx, count, **skw)
            ms_fused = sw.timeit(lambda: fused(out, pval, pidx, count))
            ms_two = sw.timeit(two_launches)
            ms_plain = sw.timeit(lambda: plain(check)) if "--parts" in sys.argv else None
            frac = lambda ms: nbytes / ms / 1e6 / 80.0
            regret = max(0.0, frac(ms_two) - frac(ms_fused))
            if regret > worst[0]:
                worst = (regret, "%s at %g" % (name, density))
            print("%-44s patches %-7g fused %5.1f   decode + scatter %5.1f   regret %4.1f  (relative %4.1f %%)  %s" % (
                name, density, frac(ms_fused), frac(ms_two), regret, 100.0 * regret / frac(ms_two), "ok" if ok else "MISMATCH") +
                ("" if ms_plain is None else "   [ms: fused %.3f, two launches %.3f, the plain decoder alone %.3f]" % (ms_fused, ms_two, ms_plain)), flush=True)
        del out, check
        torch.cuda.empty_cache()
    print("# worst regret: %.1f points (%s)" % worst)


if __name__ == "__main__":
    main()
