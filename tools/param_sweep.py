"""Performance across segment lengths / run lengths (development aid): looks for cliffs away from the BASELINE shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import sweep_all as sw
n = 1 << 28
def show(name, res):
    ms, nbytes, ok = res
    print("%-50s %8.3f ms %8.1f GB/s %5.1f%% %s" % (name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
for L in (128, 256, 512, 1024, 2048, 4096, 16384, 32768, 1 << 16, 1 << 20, 1 << 24):
    show("delta<4,8,2> L=%d" % L, sw.run_delta(4, 8, 2, n, L))
for L in (1 << 17, 1 << 20, 1 << 24, 1 << 28):
    show("delta<4,8,2> L=%d, in parts (part_anchors_scratch)" % L, sw.run_delta(4, 8, 2, n, L, in_parts=True))
for L in (256, 1024, 4096, 1 << 16):
    show("delta<4,4,1> L=%d" % L, sw.run_delta(4, 4, 1, n, L))
for mean_run in (2, 4, 8, 32, 128, 1000, 20000):
    for L in (1024, 4096, 65536):
        show("rle<4,4,2> mean_run=%d L=%d" % (mean_run, L), sw.run_rle(4, 4, 2, 4, n, L, mean_run))
for mean_run in (2, 32, 1000):
    show("rpe<4,4,2,rel> mean_run=%d L=4096" % mean_run, sw.run_rpe(4, 4, 2, True, 4, n, 4096, mean_run))
for L in (128, 256, 1000, 3000):   # short segments and segments that are not whole tiles / rows
    m = n // L * L
    show("rle<4,4,2> L=%d" % L, sw.run_rle(4, 4, 2, 4, m, L))
    show("rpe<4,4,2,rel> L=%d" % L, sw.run_rpe(4, 4, 2, True, 4, m, L))
    show("rpe<4,4,4,abs> L=%d" % L, sw.run_rpe(4, 4, 4, False, 4, m, L))
    show("delta<4,8,2> L=%d" % L, sw.run_delta(4, 8, 2, m, L))
    show("dzbv<8,8,4> L=%d" % L, sw.run_dzb_variable(8, 8, 4, (1 << 27) // (L * 16) * (L * 16), L))
for L in (128, 256, 1024, 4096, 65536):
    show("dzbv<8,8,4> 2-bit L=%d" % L, sw.run_dzb_variable(8, 8, 4, 1 << 27, L))
for L in (64, 1024, 4096, 1 << 20):
    show("for<4,i32,i16,linear> L=%d" % L, sw.run_for(4, 4, 2, "linear", 4, n, L))
