"""RLE / RPE over mean run lengths (development aid): the rows of tools/param_sweep.py the round-2 verdict asks about.
    python tools/short_runs.py [log2n=28]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import sweep_all as sw
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
tag = os.environ.get("TAG", "")
def show(name, res):
    ms, nbytes, ok = res
    print("%-10s %-46s %8.3f ms %8.1f GB/s %5.1f%% %s" % (tag, name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
for mean_run in (2, 3, 4, 6, 8, 16, 32):
    for L in (1024, 4096):
        show("rle<4,4,2> mean_run=%d L=%d" % (mean_run, L), sw.run_rle(4, 4, 2, 4, n, L, mean_run))
for mean_run in (2, 4, 8, 32):
    show("rpe<4,4,2,rel> mean_run=%d L=4096" % mean_run, sw.run_rpe(4, 4, 2, True, 4, n, 4096, mean_run))
    show("rpe<4,4,4,abs> mean_run=%d L=4096" % mean_run, sw.run_rpe(4, 4, 4, False, 4, n, 4096, mean_run))
