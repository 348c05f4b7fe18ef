"""With a library built with -DGIDDY_STREAM_STAMP_PROBE: when was each tile of a streaming decoder (Dictionary, ...) finished, and on which XCD?
How evenly do the eight XCDs get through their shares, and what does the launch lose at its end?
    GIDDY_LIBRARY_PATH=variants/sstamp.so python tools/stream_stamp_probe.py [workload=dictionary] [log2n=30]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "dictionary"
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
w = bench.make_workload(name, 1 << log2n, torch.device("cuda", 0), cap=1 << log2n)
stream = torch.cuda.current_stream().cuda_stream
for _ in range(3): w.run(stream)
torch.cuda.synchronize()
out = w.output.view(torch.int64)
# a tile = threads per workgroup x steps x 16 bytes of output; find it from where stamps (huge values with small low nibbles) recur
tile_bytes = int(os.environ.get("TILE_BYTES", 256 * 8 * 16))
stride = tile_bytes // 8
stamps = out[::stride].cpu().numpy().astype(np.uint64)
t = (stamps >> np.uint64(4)).astype(np.int64); xcc = (stamps & np.uint64(15)).astype(np.int64)
ok = (t > np.median(t) - 10**8) & (t < np.median(t) + 10**8)
print("%s: tiles %d (of %d bytes), stamps that look like stamps: %.2f%%" % (name, len(t), tile_bytes, 100 * ok.mean()))
t = t - t[ok].min(); us = t / 100.0
tile = np.arange(len(t))
print("launch lasted %.1f us; XCD of tile == tile %% 8 for %.1f%% of tiles" % (us[ok].max(), 100 * np.mean(xcc[ok] == tile[ok] % 8)))
last = [us[ok & (xcc == x)].max() for x in range(8)]
print("an XCD's last tile finished at [us]:", " ".join("%.1f" % v for v in last))
print("idle XCD-time at the end: %.1f us of %.1f = %.2f%% of the launch" % (sum(max(last) - v for v in last) / 8, max(last), 100 * sum(max(last) - v for v in last) / 8 / max(last)))
half = us[ok].max() / 2
print("tiles done per XCD by half time:", " ".join("%d" % ((us < half) & ok & (xcc == x)).sum() for x in range(8)))
