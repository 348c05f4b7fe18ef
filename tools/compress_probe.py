"""The run compressors alone, no decode (development aid: safe with builds whose output is wrong on purpose, e.g. GIDDY_CHAINED_TIMING_WITHOUT_THE_CHAIN)
    python tools/compress_probe.py [log2n=28] [mean_run ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from libgiddy_amd import compressors
import workloads as w

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << log2n
dev = torch.device("cuda", 0)
CHAIN = {"patience": 1 << 23}   # (naming the chain's arguments asks for the one-launch kernels of rounds 3-4)
for mean_run in [int(a) for a in sys.argv[2:]] or [32]:
    x = w.RunLength(n, dev, mean_run=mean_run)
    x.run(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    column = x.output.clone()
    del x; torch.cuda.empty_cache()
    capacity = n if mean_run < 8 else n // 4
    for name, fn in (("run_length_encoding<4,4,2>", lambda **how: compressors.run_length_encoding(column, 4096, np.uint16, capacity=capacity, **how)),
                     ("run_position_encoding<4,4,2,true>", lambda **how: compressors.run_position_encoding(column, 4096, np.uint16, True, capacity=capacity, **how))):
        for label, how in (("listed", {}), ("chained", {"chain": CHAIN})):
            fn(**how); torch.cuda.synchronize()
            best = float("inf")
            for _ in range(5):
                t = time.perf_counter(); f = fn(**how); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
            print("mean run %-4d %-36s %-8s %8.3f ms wall  %7.1f GB/s of column  runs %d" % (mean_run, name, label, best * 1e3, 4 * n / best / 1e9, f["num_element_runs"]), flush=True)
    del column; torch.cuda.empty_cache()
