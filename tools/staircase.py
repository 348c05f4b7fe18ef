"""development aid (GPU box): kernel time against column length — is it a staircase?  A decoder whose wavefronts take a chunk each and
whose time is set by how many ROUNDS of resident wavefronts the chunks make, not by the bytes, shows steps.
    python tools/staircase.py [I,U,C] [L] [step_log2] [steps]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import sweep_all as sw

I, U, C = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,4,2").split(","))
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
step = 1 << (int(sys.argv[3]) if len(sys.argv) > 3 else 23)
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
for k in range(1, steps + 1):
    n = k * step // L * L
    ms, nbytes, ok = sw.run_delta(I, U, C, n, L)
    print("delta<%d,%d,%d> L=%d n=%10d %8.1f us %5.1f%% of 8 TB/s  %6.2f us per 2^20 elements %s" % (I, U, C, L, n, ms * 1e3, nbytes / ms / 1e6 / 80, ms * 1e3 / (n / 2 ** 20), "" if ok else "MISMATCH"), flush=True)
