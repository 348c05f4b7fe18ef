"""python tools/alignment_probe.py [workload] — does the rate of a run decoder depend on WHERE its output lies?  (development aid, GPU box)
The C4 workload is built once; then its output is replaced by views at other offsets inside a larger allocation and the launch is timed again."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench

dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "run_length_encoding"
work = bench.make_workload(name, 1 << 30, dev)
n = work.n


def rate(tag):
    if hasattr(work, "_launch"):
        work.config = work._launch(torch.cuda.current_stream().cuda_stream, None)   # (the launch configuration does not depend on the address; resolved again all the same)
    el, ms = bench.timed_steps(work, 20, 3, lambda: None)
    print("%-28s %-50s %8.4f ms  %5.1f%%   out %x" % (name, tag, ms, work.algorithmic_bytes / ms / 1e6 / 80.0, work.output.data_ptr()), flush=True)


rate("as built"); rate("as built (again)")
big = torch.empty(n + (1 << 24), dtype=work.output.dtype, device=dev)
size = work.output.element_size()
for rep in range(2):
    for off in (0, 128, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 1 << 17, 1 << 18, 1 << 19, 1 << 20, 1 << 21, 3 << 20):   # in bytes
        work.output = big[off // size:off // size + n]
        rate("output at +%d bytes of a fresh allocation" % off)
