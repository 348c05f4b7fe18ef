"""GPU box: the device-side compressors on columns at the EDGES — one value throughout, no two neighbours equal, runs longer than a length field holds, one patch of
wide values in a narrow column — each compressed, decoded back and compared (2^26 elements): where do the chains or the hash set fall over?   python tools/compress_edges.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from libgiddy_amd import compressors, schemes

n = 1 << int(os.environ.get("LOG2N", "26"))
dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    best = float("inf")
    for _ in range(reps):
        t = time.perf_counter(); out = fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t)
    return out, best

def line(name, seconds, ok, extra=""):
    print("%-66s %8.2f ms  %7.1f GB/s of column  %s %s" % (name, seconds * 1e3, 4 * n / seconds / 1e9, "ok" if ok else "MISMATCH", extra), flush=True)

columns = {
    "one value throughout": torch.full((n,), 12345, dtype=torch.int32, device=dev),
    "no two neighbours equal": (torch.arange(n, device=dev, dtype=torch.int32) & 1023),
    "runs of 100 000": (torch.arange(n, device=dev, dtype=torch.int64) // 100000).to(torch.int32),
    "mean run 32 with one run of 2^24": None,
}
g = torch.Generator(device=dev); g.manual_seed(5)
mixed = torch.repeat_interleave(torch.randint(0, 1 << 20, (n // 32 + 1,), dtype=torch.int32, device=dev, generator=g), 32)[:n].clone()
mixed[n // 3: n // 3 + (1 << 24)] = 7
columns["mean run 32 with one run of 2^24"] = mixed

for name, column in columns.items():
    try:
        f, t = timed(lambda: compressors.run_length_encoding(column, 4096, np.uint16, capacity=n))
        out = torch.empty_like(column)
        schemes.run_length_encoding(out, f["run_data"], f["run_lengths"], f["position_anchors"], f["intra_run_anchor_offsets"], 4096, f["num_segments"], f["num_element_runs"], n, stream=st)
        line("run lengths (16-bit) <- " + name, t, bool((out == column).all()), "%d pieces" % f["num_element_runs"])
    except Exception as e:
        print("run lengths <- %s: %s" % (name, str(e).split(chr(10))[0][:120]), flush=True)
    try:
        f, t = timed(lambda: compressors.run_position_encoding(column, 4096, np.uint16, True, capacity=n))
        out = torch.zeros_like(column)
        schemes.run_position_encoding(out, f["run_data"], f["run_start_positions"], f["position_anchors"], 4096, f["num_anchors"], f["num_element_runs"], n, stream=st)
        line("run positions (relative, 16-bit) <- " + name, t, bool((out == column).all()), "%d runs" % f["num_element_runs"])
    except Exception as e:
        print("run positions <- %s: %s" % (name, str(e).split(chr(10))[0][:120]), flush=True)
