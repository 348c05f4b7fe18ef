"""development aid (GPU box): the stand-alone scatter kernel on ascending positions at patch densities of 5-50 % of a 2^28-element column — what "decode, then scatter" would cost
where the fused ManyPatches builds are slow.   python tools/scatter_dense.py"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import sweep_all as sw
from libgiddy_amd import schemes
n = 1 << 28
st = torch.cuda.current_stream().cuda_stream
for E in (4, 8):
    target = torch.zeros(n, dtype=sw.INT[E], device=sw.DEV)
    for fraction in (0.001, 0.05, 0.2, 0.5):
        g = sw.gen(int(fraction * 1000) + E)
        p = int(n * fraction)
        idx = torch.sort(torch.randperm(n, generator=g, device=sw.DEV)[:p])[0].to(torch.int32)
        val = sw.rand_bits(p, E, g)
        kw = dict(output_index_size=4, element_size=E, input_index_size=4, stream=st)
        cfg = schemes.scatter(target, val, idx, p, **kw)
        ms = sw.timeit(lambda: schemes.scatter(target, val, idx, p, launch_config=cfg, **kw))
        print("scatter of %d-byte values, %5.1f %% of 2^28 positions: %8.3f ms  (%.0f GB/s of positions + values + written elements)" % (E, fraction * 100, ms, p * (4 + 2 * E) / ms / 1e6), flush=True)
        del idx, val
