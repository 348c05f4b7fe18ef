"""development aid (GPU box): Delta at whole-row segments with the OUTPUT (and input) placed at byte offsets from a 2 MiB boundary — does a
wavefront's 32 KiB chunk have to sit on a 32 KiB boundary, or only be 32 KiB long?      python tools/shifted_bases.py [I,U,C] [L]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import sweep_all as sw
from libgiddy_amd import schemes

I, U, C = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,4,2").split(","))
L = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n = (1 << 28) // L * L
g = sw.gen(1)
d_all = sw.rand_bits(n + (1 << 20), C, g)
out_all = torch.empty(n + (1 << 20), dtype=sw.INT[U], device=sw.DEV)
st = torch.cuda.current_stream().cuda_stream
for out_shift, in_shift in ((0, 0), (1024, 0), (4096, 0), (16384, 0), (16384 + 1024, 0), (0, 512), (0, 8192), (16384, 8192), (32768, 16384), (1024, 512)):
    out = out_all[out_shift // U: out_shift // U + n]
    d = d_all[in_shift // C: in_shift // C + n]
    du = sw.unsigned(d, C)
    per_seg = du.view(n // L, L).sum(dim=1)
    anchors = ((1 << 40) + torch.cumsum(per_seg, 0) - per_seg).to(sw.INT[U])
    kw = dict(uncompressed_size=U, compressed_size=C, index_size=I, stream=st)
    cfg = schemes.delta(out, d, anchors, n, L, **kw)
    ms = sw.timeit(lambda: schemes.delta(out, d, anchors, n, L, launch_config=cfg, **kw))
    print("delta<%d,%d,%d> L=%d out +%6d B, in +%6d B: %7.1f us %5.1f%%" % (I, U, C, L, out_shift, in_shift, ms * 1e3, ((U + C) * n + U * (n // L)) / ms / 1e6 / 80), flush=True)
