import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import sweep_all as sw
from libgiddy_amd import schemes, registered as r
p = torch.cuda.get_device_properties(0)
print("props:", p.multi_processor_count, getattr(p, "shared_memory_per_multiprocessor", None), getattr(p, "shared_memory_per_block", None), getattr(p, "shared_memory_per_block_optin", None))
n = 1 << 28
for U, D, entries_n in ((8, 4, 4096), (4, 4, 4096), (8, 8, 4096), (8, 2, 4096)):
    g = sw.gen(1)
    entries = sw.rand_bits(entries_n, U, g); keys = sw.rand_bits(n, D, g, high=entries_n)
    out = torch.empty(n, dtype=sw.INT[U], device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for cache in (True, False):
        for grid in (256, 512, 1024, 1280, 2048, 4096, 16384, None):
            lim = r.launch_configuration_limits_t(grid_size=grid) if grid else None
            try:
                cfg = schemes.dictionary(out, keys, entries, n, entries_n, uncompressed_size=U, dictionary_index_size=D, stream=st, limits=lim, cache_dictionary_in_shared_mem=cache)
            except Exception as e:
                print("fail", e); continue
            ms = sw.timeit(lambda: schemes.dictionary(out, keys, entries, n, entries_n, uncompressed_size=U, dictionary_index_size=D, stream=st, launch_config=cfg, cache_dictionary_in_shared_mem=cache))
            print("U=%d D=%d cache=%s grid_limit=%s -> cfg %s: %.3f ms %.0f GB/s" % (U, D, cache, grid, cfg.grid_dimensions[0], ms, (U + D) * n / ms / 1e6))
