"""Development probe: does LIMITING occupancy (by padding the dynamic LDS of a launch) speed the streaming decoders up?
    python tools/occupancy_probe.py"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench
from libgiddy_amd import schemes

n = 1 << 30
dev = torch.device("cuda", 0)
for name in ("dictionary", "frame_of_reference", "discard_zero_bytes_fixed", "model"):
    w = bench.make_workload(name, n, dev)
    base = w.config.dynamic_shared_memory_size
    for lds_kib in (0, 18, 22, 26, 31, 39, 52):
        w.config.dynamic_shared_memory_size = max(base, lds_kib * 1024)
        try:
            el, ms = bench.timed_steps(w, 10, 2, lambda: None)
        except Exception as e:
            print(name, lds_kib, "failed:", e); continue
        per_cu = 160 // max(1, (w.config.dynamic_shared_memory_size + 1023) // 1024)
        print("%-28s dynamic LDS %2d KiB (<= %2d workgroups per CU): %.4f ms  %.0f GB/s algorithmic (%.1f %%)" % (
            name, w.config.dynamic_shared_memory_size // 1024, min(8, per_cu), ms, w.algorithmic_bytes / ms / 1e6, w.algorithmic_bytes / ms / 1e6 / 80), flush=True)
    del w
    torch.cuda.empty_cache()
