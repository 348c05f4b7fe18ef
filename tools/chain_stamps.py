"""Where a chained compressor's tiles spend their time (development aid; needs a build with -DGIDDY_CHAINED_STAMPS)
    GIDDY_LIBRARY_PATH=variants/stamps.so python tools/chain_stamps.py [log2n=28]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from libgiddy_amd import compressors
import workloads as w
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << log2n
dev = torch.device("cuda", 0)
x = w.RunLength(n, dev); x.run(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
column = x.output.clone(); del x
for _ in range(2):
    compressors.run_position_encoding(column, 4096, np.uint16, True, capacity=n // 8); torch.cuda.synchronize()
tiles = n // 16384
s = compressors.LAST_SCRATCH.cpu().numpy()[tiles: 5 * tiles].reshape(tiles, 4).astype(np.int64)
t0 = s[:, 0].min()
s = (s - t0) / 100.0   # microseconds
print("tiles %d; kernel span %.1f us" % (tiles, s[:, 3].max()))
for name, a, b in (("load + count", 0, 1), ("chain", 1, 2), ("write", 2, 3), ("whole", 0, 3)):
    d = s[:, b] - s[:, a]
    print("%-14s mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  p99 %7.2f  max %7.2f us" % (name, d.mean(), *np.percentile(d, [10, 50, 90, 99]), d.max()))
order = np.argsort(s[:, 0])
print("start order vs tile order: mean |rank - tile| = %.1f" % np.abs(np.argsort(order) - np.arange(tiles)).mean())
for k in (0, 100, 1000, 2000, 4000, 8000, 16000):
    if k < tiles: print("tile %6d: start %8.2f counted %8.2f based %8.2f done %8.2f" % (k, *s[k]))
# how many tiles are between start and end at a time
starts = np.sort(s[:, 0]); ends = np.sort(s[:, 3])
for at in (50, 100, 200, 300, 400):
    print("at %4d us: started %6d finished %6d in flight %5d" % (at, np.searchsorted(starts, at), np.searchsorted(ends, at), np.searchsorted(starts, at) - np.searchsorted(ends, at)))
