import sys, os; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import torch
import sweep_all as sw
def show(name, res):
    ms, nbytes, ok = res
    print("%-60s %8.3f ms %8.1f GB/s %5.1f%% %s" % (name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
for L in (1000, 1536, 3000, 5000, 10000, 4096):
    n = (1 << 29) // L * L
    show("delta <4,8,2> L=%d" % L, sw.run_delta(4, 8, 2, n, L)); torch.cuda.empty_cache()
for L in (1000, 3000, 4096):
    n = (1 << 29) // L * L
    show("delta <4,4,1> L=%d" % L, sw.run_delta(4, 4, 1, n, L)); torch.cuda.empty_cache()
for L in (1000, 3000, 10000, 4096):
    n = (1 << 29) // L * L
    show("rle <4,4,2,4> L=%d" % L, sw.run_rle(4, 4, 2, 4, n, L)); torch.cuda.empty_cache()
    show("for <4,int,short,linear> L=%d" % L, sw.run_for(4, 4, 2, "linear", 4, n, L)); torch.cuda.empty_cache()
for L in (1000, 1536, 3000, 1024):
    n = (1 << 27) // (L * 16) * (L * 16)
    show("dzbv <8,8,4> L=%d" % L, sw.run_dzb_variable(8, 8, 4, n, L)); torch.cuda.empty_cache()
