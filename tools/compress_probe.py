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
x = w.RunLength(n, dev)
x.run(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
column = x.output.clone()
del x; torch.cuda.empty_cache()
for name, fn in (("run_length_encoding<4,4,2>", lambda: compressors.run_length_encoding(column, 4096, np.uint16, capacity=n // 8)),
                 ("run_position_encoding<4,4,2,true>", lambda: compressors.run_position_encoding(column, 4096, np.uint16, True, capacity=n // 8))):
    fn(); torch.cuda.synchronize()
    best = float("inf")
    for _ in range(5):
        t = time.perf_counter(); f = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    print("%-40s %8.3f ms wall  %7.1f GB/s of column  runs %d" % (name, best * 1e3, 4 * n / best / 1e9, f["num_element_runs"]), flush=True)
