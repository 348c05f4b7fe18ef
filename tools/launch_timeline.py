"""Where a launch of a streaming decoder spends its time: with a library built with -DGIDDY_STREAM_STAMP_PROBE every tile writes WHEN it finished
(s_memrealtime, 100 MHz) and on which XCD in place of its first 8 bytes of output.
    GIDDY_LIBRARY_PATH=variants/sstamp.so python tools/launch_timeline.py WORKLOAD LOG2N [TILE_BYTES]
Prints, for one launch in the middle of a train of launches over rotated (cold) columns: the launch's duration by events, the span of its stamps,
how many tiles each XCD had finished by every 5 us, and when each XCD finished its first and its last tile — the ramp, the paces and the tail."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import bench
name = sys.argv[1]; log2n = int(sys.argv[2])
tile_bytes = int(sys.argv[3]) if len(sys.argv) > 3 else 256 * 4 * 16
device = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
columns = [bench.make_workload(name, 1 << log2n, device, cap=1 << log2n, seed_offset=1000 * i) for i in range(max(3, (3 << 29) // (5 << log2n)))][:12]
configs = [w.launch_fields(w.output, w.column_fields(), stream, None) for w in columns]
torch.cuda.synchronize()
events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in columns]
for _ in range(2):
    for w, c, (a, b) in zip(columns, configs, events):
        a.record(); w.launch_fields(w.output, w.column_fields(), stream, c); b.record()
torch.cuda.synchronize()
def compact(index):
    w = columns[index]
    stamps = w.output.view(torch.int64)[::tile_bytes // 8].cpu().numpy().astype(np.uint64)
    t = (stamps >> np.uint64(4)).astype(np.int64); xcc = (stamps & np.uint64(15)).astype(np.int64)
    ok = (t > np.median(t) - 10**7) & (t < np.median(t) + 10**7) & (xcc < 8)
    us = (t - t[ok].min()) / 100.0
    return "%.1f us by events; XCDs' last tiles at " % (events[index][0].elapsed_time(events[index][1]) * 1e3) + " ".join("%.0f" % us[ok & (xcc == x)].max() for x in range(8))
for index in range(len(columns)):
    print("   column %d (output at 0x%x): %s" % (index, columns[index].output.data_ptr(), compact(index)))
for index in (len(columns) // 2,):
    w = columns[index]
    ms = events[index][0].elapsed_time(events[index][1])
    out = w.output.view(torch.int64)
    stamps = out[::tile_bytes // 8].cpu().numpy().astype(np.uint64)
    t = (stamps >> np.uint64(4)).astype(np.int64); xcc = (stamps & np.uint64(15)).astype(np.int64)
    ok = (t > np.median(t) - 10**7) & (t < np.median(t) + 10**7) & (xcc < 8)
    us = (t - t[ok].min()) / 100.0
    span = us[ok].max()
    print("%s 2^%d: %d tiles of %d bytes (%.2f %% carry a stamp); one launch: %.1f us by events, its stamps span %.1f us -> %.1f us before the first tile finished + after the last"
          % (name, log2n, len(t), tile_bytes, 100 * ok.mean(), ms * 1e3, span, ms * 1e3 - span))
    print("   an XCD's first / last tile finished at [us]: " + "  ".join("%.1f/%.1f" % (us[ok & (xcc == x)].min(), us[ok & (xcc == x)].max()) for x in range(8)))
    last = [us[ok & (xcc == x)].max() for x in range(8)]
    print("   idle XCD-time at the end: %.1f us average (%.1f %% of the launch); tiles per XCD: %s" % (sum(max(last) - v for v in last) / 8, 100 * sum(max(last) - v for v in last) / 8 / max(span, 1e-9),
          " ".join("%d" % (ok & (xcc == x)).sum() for x in range(8))))
    step = 5.0 if span < 200 else span / 40
    edges = np.arange(0, span + step, step)
    print("   tiles finished per %.1f us, all XCDs (%% of the launch's best bin):" % step)
    counts = np.histogram(us[ok], bins=edges)[0]
    print("   " + " ".join("%3d" % round(100 * c / max(counts.max(), 1)) for c in counts))
