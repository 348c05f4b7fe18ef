"""Run-length columns in long segments by mean run: the pre-pass that finds finer anchors as ONE launch against the three
launches of round 3 (GIDDY_RUN_FINE_ONE_LAUNCH=0) — every figure checked bit for bit.
    TAG=one_launch python tools/fine_prepass_sweep.py [log2n=28]; GIDDY_RUN_FINE_ONE_LAUNCH=0 TAG=three_launches python tools/fine_prepass_sweep.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sweep_all as sw
log2n = ([int(a) for a in sys.argv[1:] if a.isdigit()] or [28])[0]
n = 1 << log2n
tag = os.environ.get("TAG", "")
def show(name, res):
    ms, nbytes, ok = res
    print("%-16s %-46s %8.3f ms %8.1f GB/s %5.1f%% %s" % (tag, name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
MEANS = [int(x) for x in os.environ.get("MEANS", "2,4,8,32,1000").split(",")]
LS = [int(x) for x in os.environ.get("LS", "4096,16384,65536,1048576").split(",")]
for mean_run in MEANS:
    for L in LS:
        show("rle<4,4,2> 2^%d mean_run=%d L=%d" % (log2n, mean_run, L), sw.run_rle(4, 4, 2, 4, n, L, mean_run))
for mean_run in ((4, 32) if "OTHERS" in os.environ else ()):
    for L in (65536,):
        show("rle<4,1,1> 2^%d mean_run=%d L=%d" % (log2n, mean_run, L), sw.run_rle(4, 1, 1, 4, n, L, mean_run))
        show("rle<4,8,4> 2^%d mean_run=%d L=%d" % (log2n, mean_run, L), sw.run_rle(4, 8, 4, 4, n, L, mean_run))
        show("rle<8,8,8,8> 2^%d mean_run=%d L=%d" % (log2n, mean_run, L), sw.run_rle(8, 8, 8, 8, n, L, mean_run))
