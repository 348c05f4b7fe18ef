"""Why does a decoder lose points on a bigger column?  One workload at 2^log2n:  (a) one launch;  (b) the same column as PIECES launches of
consecutive shards, back to back;  (c) single shards of it alone;  (d) the timing loop's length (sustained clocks).
    python tools/size_probe.py WORKLOAD [log2n=31] [pieces=8]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench

name = sys.argv[1]; log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 31; pieces = int(sys.argv[3]) if len(sys.argv) > 3 else 8
device = torch.device("cuda", 0)
w = bench.make_workload(name, 1 << log2n, device, cap=1 << log2n)
stream = torch.cuda.current_stream().cuda_stream


def timed(launch, steps):
    for _ in range(2): launch()
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(steps): launch()
    stop.record(); torch.cuda.synchronize()
    return start.elapsed_time(stop) / steps


def pct(nbytes, ms): return nbytes / ms / 1e6 / 80

whole_config = w.launch_fields(w.output, w.column_fields(), stream, None)
for steps in (5, 20, 80):
    ms = timed(lambda: w.launch_fields(w.output, w.column_fields(), stream, whole_config), steps)   # (the decoder alone: no patch launch behind Delta)
    print("%s 2^%d, one launch, %3d steps in a row: %.4f ms = %.1f%%" % (name, log2n, steps, ms, pct(w.algorithmic_bytes, ms)), flush=True)

fields = w.column_fields()
shards = [w.shard_function(fields, r, pieces) for r in range(pieces)]
outs = [w.output[s.start:s.stop] for s in shards]
configs = [w.launch_fields(o, s.fields, stream, None) for o, s in zip(outs, shards)]


def all_pieces():
    for o, s, c in zip(outs, shards, configs): w.launch_fields(o, s.fields, stream, c)

ms = timed(all_pieces, 10)
print("%s 2^%d as %d launches of consecutive shards: %.4f ms = %.1f%%" % (name, log2n, pieces, ms, pct(w.algorithmic_bytes, ms)), flush=True)
for r in (0, pieces // 2, pieces - 1):
    ms = timed(lambda: w.launch_fields(outs[r], shards[r].fields, stream, configs[r]), 20)
    print("   shard %d alone (1/%d of the bytes): %.4f ms = %.1f%%" % (r, pieces, ms, pct(w.algorithmic_bytes / pieces, ms)), flush=True)
if hasattr(w, "values"):
    assert bool((w.output == w.values).all())
