"""Decoders over LONG segments (development aid): the rows of tools/param_sweep.py that the round-2 verdict asks about, each checked bit for bit.
    python tools/long_segments.py [delta] [rle] [rpe] [dzbv] [log2n=28]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import sweep_all as sw
which = [a for a in sys.argv[1:] if not a.isdigit()] or ["delta", "rle", "rpe", "dzbv"]
log2n = ([int(a) for a in sys.argv[1:] if a.isdigit()] or [28])[0]
n = 1 << log2n
tag = os.environ.get("TAG", "")
def show(name, res):
    ms, nbytes, ok = res
    print("%-10s %-46s %8.3f ms %8.1f GB/s %5.1f%% %s" % (tag, name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
LONG = (4096, 8192, 16384, 32768, 1 << 16, 1 << 20)
if "delta" in which:
    for L in LONG + (1 << 24, n):
        show("delta<4,8,2> L=%d" % L, sw.run_delta(4, 8, 2, n, L))
    for L in (16384, 1 << 16, 1 << 20, n):
        show("delta<4,4,1> L=%d" % L, sw.run_delta(4, 4, 1, n, L))
        show("delta<8,8,4> L=%d" % L, sw.run_delta(8, 8, 4, n, L))
if "rle" in which:
    for mean_run in (32, 4):
        for L in LONG:
            show("rle<4,4,2> mean_run=%d L=%d" % (mean_run, L), sw.run_rle(4, 4, 2, 4, n, L, mean_run))
if "rpe" in which:
    for L in LONG[:5]:
        show("rpe<4,4,2,rel> L=%d" % L, sw.run_rpe(4, 4, 2, True, 4, n, L, 32))
    for L in LONG:
        show("rpe<4,4,4,abs> L=%d" % L, sw.run_rpe(4, 4, 4, False, 4, n, L, 32))
if "dzbv" in which:
    for L in (1024,) + LONG:
        show("dzbv<8,8,4> 2-bit L=%d" % L, sw.run_dzb_variable(8, 8, 4, min(n, 1 << 28), L))
