"""development aid (GPU box): a grid-stride decoder under grids of different sizes (the kernels take any grid): covering — a chunk per
wavefront, the hardware dispatcher hands out workgroups as others retire — against grids the size of what is resident at once.
    python tools/grid_sizes.py [I,U,C] [L ...]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import sweep_all as sw
from libgiddy_amd import schemes, registered as r

I, U, C = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "4,4,2").split(","))
for L in [int(x) for x in sys.argv[2:]] or [48, 64]:
    n = (1 << 28) // L * L
    d = sw.rand_bits(n, C, sw.gen(1))
    out = torch.empty(n, dtype=sw.INT[U], device=sw.DEV)
    du = sw.unsigned(d, C)
    per_seg = du.view(n // L, L).sum(dim=1)
    anchors = ((1 << 40) + torch.cumsum(per_seg, 0) - per_seg).to(sw.INT[U])
    st = torch.cuda.current_stream().cuda_stream
    kw = dict(uncompressed_size=U, compressed_size=C, index_size=I, stream=st)
    cfg = schemes.delta(out, d, anchors, n, L, **kw)
    covering = cfg.grid_dimensions[0]
    for grid in [covering] + [256 * k for k in (2, 4, 5, 6, 7, 8, 9, 10, 12, 16)] + [covering // 2, covering // 3, covering // 4]:
        custom = r.launch_configuration_t((grid, 1, 1), cfg.block_dimensions, cfg.dynamic_shared_memory_size)
        ms = sw.timeit(lambda: schemes.delta(out, d, anchors, n, L, launch_config=custom, **kw))
        print("delta<%d,%d,%d> L=%-5d grid %6d x %d (covering %d): %7.1f us %5.1f%%" % (I, U, C, L, grid, cfg.block_dimensions[0], covering, ms * 1e3, ((U + C) * n + U * (n // L)) / ms / 1e6 / 80), flush=True)
