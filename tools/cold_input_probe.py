"""One column repeated against three columns in turn (the input then comes from HBM, not from the Infinity Cache) for a workload:
    python tools/cold_input_probe.py WORKLOAD [log2n=30]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench
name = sys.argv[1]; log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
device = torch.device("cuda", 0)
w = bench.make_workload(name, 1 << log2n, device, cap=1 << log2n)
w.run(torch.cuda.current_stream().cuda_stream)
from oracle import oracle
w.verify(oracle)
for rep in range(2):
    r = bench.placement_and_cache(name, w, 1 << log2n, device, 12)
    print(os.path.basename(os.environ.get("GIDDY_LIBRARY_PATH", "default")), name, r, flush=True)
