import sys, os; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import torch
import sweep_all as sw
def show(name, res):
    ms, nbytes, ok = res
    print("%-60s %8.3f ms %8.1f GB/s %5.1f%% %s" % (name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
n = 1 << 30
for U, C in ((8, 2), (4, 1), (8, 4), (4, 2), (2, 1)):
    show("delta <4,%d,%d> 2^30" % (U, C), sw.run_delta(4, U, C, n)); torch.cuda.empty_cache()
