"""With a library built with -DGIDDY_DZBV_STAMP_PROBE: when was each DZB-V segment finished, and on which XCD?  Throughput over the launch's
lifetime and how far apart the XCDs' fronts are.     GIDDY_LIBRARY_PATH=variants/k1stamp.so python tools/stamp_probe.py [log2n=30] [bins=24]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import bench

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30; bins = int(sys.argv[2]) if len(sys.argv) > 2 else 24
w = bench.make_workload("discard_zero_bytes_variable", 1 << log2n, torch.device("cuda", 0), cap=1 << log2n)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3): w.run(stream)
torch.cuda.synchronize()
stamps = w.output[:: w.L].cpu().numpy().astype(np.uint64)
t = (stamps >> np.uint64(4)).astype(np.int64); xcc = (stamps & np.uint64(15)).astype(np.int64)
t -= t.min(); t_us = t / 100.0
seg = np.arange(len(t)); block = seg // 4
print("segments %d, launch lasted %.1f us; XCD of block b == b %% 8 for %.2f%% of blocks" % (len(t), t_us.max(), 100.0 * np.mean(xcc == block % 8)))
edges = np.linspace(0, t_us.max() + 1e-9, bins + 1)
which = np.digitize(t_us, edges) - 1
bytes_per_segment = w.algorithmic_bytes / len(t)
print("%12s %10s %8s | front (median finished segment / total) per XCD ... | spread of the XCDs' fronts (segments, MiB of output)" % ("t [us]", "segments", "% peak"))
for b in range(bins):
    m = which == b
    width = edges[b + 1] - edges[b]
    rate = m.sum() * bytes_per_segment / (width * 1e-6) / 8e12 * 100
    fronts = [np.median(seg[m & (xcc == x)]) if (m & (xcc == x)).any() else np.nan for x in range(8)]
    spread = np.nanmax(fronts) - np.nanmin(fronts)
    inflight = np.percentile(seg[m], 99) - np.percentile(seg[m], 1) if m.any() else 0
    print("%12.1f %10d %8.1f | %s | %8.0f = %6.1f MiB ; p1-p99 of segments finished in this bin spans %.0f MiB" % (
        edges[b + 1], m.sum(), rate, " ".join("%.4f" % (f / len(t)) for f in fronts), spread, spread * w.L * 8 / 2**20, inflight * w.L * 8 / 2**20))
