"""Does a timing loop over ONE column flatter a decoder?  MI355X keeps 256 MiB of Infinity Cache in front of HBM: launches that re-read the
same input find part of it there.  Times a workload (a) on one column, as bench.py does, (b) round-robin over COPIES distinct columns whose
inputs together exceed the cache several times.      python tools/mall_probe.py WORKLOAD [log2n=28] [copies=8] [steps=24]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench, workloads

name = sys.argv[1]; log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 28
copies = int(sys.argv[3]) if len(sys.argv) > 3 else 8; steps = int(sys.argv[4]) if len(sys.argv) > 4 else 24
device = torch.device("cuda", 0)
works = []
for c in range(copies):
    os.environ["RANK"] = str(c)          # make_workload derives its seed offset from the rank: distinct columns
    works.append(bench.make_workload(name, 1 << log2n, device, cap=1 << log2n))
stream = torch.cuda.current_stream().cuda_stream


def timed(sequence):
    for w in sequence[:2]: w.run(stream)
    torch.cuda.synchronize()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for w in sequence: w.run(stream)
    stop.record(); torch.cuda.synchronize()
    return start.elapsed_time(stop) / len(sequence)


w0 = works[0]
for rep in range(2):
    one = timed([w0] * steps)
    many = timed([works[i % copies] for i in range(steps)])
    print("%-30s 2^%d  one column %.4f ms = %.1f%% of 8 TB/s   |  %d columns in turn %.4f ms = %.1f%%" % (
        name, log2n, one, w0.algorithmic_bytes / one / 1e6 / 80, copies, many, w0.algorithmic_bytes / many / 1e6 / 80))
