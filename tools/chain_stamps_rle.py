"""Where the chained run-LENGTH compressor's tiles spend their time (development aid; needs a build with -DGIDDY_CHAINED_STAMPS)
    GIDDY_LIBRARY_PATH=variants/stamps.so python tools/chain_stamps_rle.py [log2n=28]"""
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
    compressors.run_length_encoding(column, 4096, np.uint16, capacity=n // 8); torch.cuda.synchronize()
tiles = n // 16384
s = compressors.LAST_SCRATCH.cpu().numpy()[2 * tiles: 8 * tiles].reshape(tiles, 6).astype(np.int64)
s = (s - s[:, 0].min()) / 100.0   # microseconds
print("tiles %d; kernel span %.1f us" % (tiles, s[:, 5].max()))
names = ("load + starts + row scans", "what reaches in (first chain)", "ends + ranks", "base (second chain)", "write")
for k, name in enumerate(names):
    d = s[:, k + 1] - s[:, k]
    print("%-32s mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  p99 %7.2f us" % (name, d.mean(), *np.percentile(d, [10, 50, 90, 99])))
d = s[:, 5] - s[:, 0]
print("%-32s mean %7.2f  p10 %7.2f  p50 %7.2f  p90 %7.2f  p99 %7.2f us" % ("whole", d.mean(), *np.percentile(d, [10, 50, 90, 99])))
