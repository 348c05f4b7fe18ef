"""A run-length column whose halves differ: mean runs of 2 in the first half, of 200 in the second.  The wrappers
pick ONE build of the decoder per column by its GLOBAL mean run (kernel_wrappers/decompression/run_length_encoding.hpp); here: the whole
column at that choice, against its halves decoded as columns of their own (each with the build its own mean run picks) — what a choice per
half would give.  Checked bit for bit.
    python tools/mixed_density.py [log2n=28]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import sweep_all as sw
from libgiddy_amd import schemes
from workloads import output_buffer

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n, L, DEV = 1 << log2n, 4096, "cuda"
g = sw.gen(77)
first, _ = sw._runs(n // 2, 2, g, 65535)
second, _ = sw._runs(n // 2, 200, g, 65535)


def column(lengths, n):
    starts = torch.cumsum(lengths, 0) - lengths
    count = len(lengths)
    values = (torch.arange(count, device=DEV) * 2654435761 % (1 << 31)).to(torch.int32)
    seg = torch.arange(0, n, L, device=DEV)
    anchor = torch.searchsorted(starts, seg, right=True) - 1
    return dict(values=values, lengths=lengths.to(torch.int16), anchors=anchor.to(torch.int32), intra=(seg - starts[anchor]).to(torch.int16), count=count, segments=len(seg), starts=starts, n=n)


def timed(c, out):
    st = torch.cuda.current_stream().cuda_stream
    args = (c["values"], c["lengths"], c["anchors"], c["intra"], L, c["segments"], c["count"], c["n"])
    cfg = schemes.run_length_encoding(out, *args, stream=st)
    ms = sw.timeit(lambda: schemes.run_length_encoding(out, *args, launch_config=cfg, stream=st), reps=15)
    ok = all(bool((c["values"][torch.searchsorted(c["starts"], torch.arange(lo, hi, device=DEV), right=True) - 1] == out[lo:hi]).all()) for lo, hi in sw.chunks(c["n"]))
    nbytes = 4 * c["n"] + 6 * c["count"] + 6 * c["segments"]
    return ms, nbytes, ok


out = output_buffer(n, torch.int32, DEV)
whole = column(torch.cat([first, second]), n)
halves = [column(first, n // 2), column(second, n // 2)]
print("run_length_encoding<4,4,2>, 2^%d outputs in segments of %d: first half mean run 2, second half mean run 200 (global mean %.1f)" % (log2n, L, n / whole["count"]))
ms, nbytes, ok = timed(whole, out)
print("  the whole column, ONE launch (the build its global mean run picks)   %.3f ms  %5.1f %% of peak  %s" % (ms, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"))
total_ms = 0
for name, c, view in (("first half alone (mean run 2: the dense build)", halves[0], out[: n // 2]), ("second half alone (mean run 200: the sparse build)", halves[1], out[n // 2:])):
    m, b, ok = timed(c, view)
    total_ms += m
    print("  %-68s %.3f ms  %5.1f %% of peak  %s" % (name, m, b / m / 1e6 / 80, "ok" if ok else "MISMATCH"))
print("  the halves one after the other: %.3f ms = %5.1f %% of peak for the column's bytes — what a choice of build per half would give" % (total_ms, nbytes / total_ms / 1e6 / 80))
