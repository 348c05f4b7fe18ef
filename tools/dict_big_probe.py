import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import sweep_all as sw
from libgiddy_amd import schemes, registered as r
n = 1 << 28
for U, D, entries_n in ((2, 2, 65536), (4, 2, 32768), (4, 2, 65536), (8, 2, 16384), (1, 1, 256), (2, 2, 16384)):
    g = sw.gen(1)
    entries = sw.rand_bits(entries_n, U, g); keys = sw.rand_bits(n, D, g, high=entries_n)
    out = torch.empty(n, dtype=sw.INT[U], device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for cache in (None, False):
        try:
            cfg = schemes.dictionary(out, keys, entries, n, entries_n, uncompressed_size=U, dictionary_index_size=D, stream=st, cache_dictionary_in_shared_mem=cache)
            ms = sw.timeit(lambda: schemes.dictionary(out, keys, entries, n, entries_n, uncompressed_size=U, dictionary_index_size=D, stream=st, launch_config=cfg, cache_dictionary_in_shared_mem=cache))
            ok = bool((entries[sw.unsigned(keys[:1 << 24], D)] == out[:1 << 24]).all())
            print("U=%d D=%d entries=%d (%d KiB) cache=%s -> %s: %.3f ms %.0f GB/s ok=%s" % (U, D, entries_n, entries_n * U >> 10, cache, cfg, ms, (U + D) * n / ms / 1e6, ok))
        except Exception as e:
            print("U=%d D=%d entries=%d cache=%s FAILED: %s" % (U, D, entries_n, cache, e))
