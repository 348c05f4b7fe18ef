"""Is the fused Dictionary kernel's shape (wavefront chunks, `sorted_patches::cursor_t` with NO patches) faster than the plain kernel's tiles on the headline column?
    python tools/experiments/dictionary_plain_vs_fused.py [log2n=30]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import workloads as w
from libgiddy_amd import schemes
log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << log2n
device = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
work = w.Dictionary(n, device)
no_values = torch.zeros(16, dtype=torch.int32, device=device)
no_positions = torch.zeros(16, dtype=torch.int32, device=device)


def timed(run, label):
    for _ in range(5): run()
    torch.cuda.synchronize()
    times = []
    for _ in range(9):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10): run()
        b.record(); torch.cuda.synchronize()
        times.append(a.elapsed_time(b) / 10)
    ms = sorted(times)[len(times) // 2]
    print("%-44s %.4f ms  %.1f %% of the HBM peak" % (label, ms, work.algorithmic_bytes / ms / 8e7), flush=True)


for turn in range(3):
    timed(lambda: work.run(stream), "plain dictionary<4,4,1>")
    config = schemes.patched_dictionary(work.output, work.keys, work.entries, no_values, no_positions, n, 256, 0, stream=stream)
    timed(lambda: schemes.patched_dictionary(work.output, work.keys, work.entries, no_values, no_positions, n, 256, 0, stream=stream, launch_config=config), "fused kernel, no patches")
work.run(stream); torch.cuda.synchronize(); reference = w.checksum(work.output)
work.output.zero_()
schemes.patched_dictionary(work.output, work.keys, work.entries, no_values, no_positions, n, 256, 0, stream=stream); torch.cuda.synchronize()
print("same output:", w.checksum(work.output) == reference)
