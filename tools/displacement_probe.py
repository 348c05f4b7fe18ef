"""GPU box: Dictionary <4,4,1> at 2^30 with the output and the keys displaced by whole 16-byte groups (the ordinary kernel where the output still starts a 128-byte
line, the any-alignment build elsewhere): does WHERE the tiles lie against the lines / pages matter?     python tools/displacement_probe.py [log2 n]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import sweep_all as sw
from libgiddy_amd import schemes

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 30)
st = torch.cuda.current_stream().cuda_stream
g = sw.gen(1)
entries = sw.rand_bits(256, 4, g); keys_all = sw.rand_bits(n + 65536, 1, g, high=256); out_all = torch.empty(n + 65536, dtype=torch.int32, device=sw.DEV)
print("keys at %#x, output at %#x" % (keys_all.data_ptr(), out_all.data_ptr()))
cases = [(0, 0), (32, 0), (0, 32), (32, 32), (64, 64), (96, 96), (128, 128), (256, 256), (1024, 1024), (4096, 4096), (0, 128), (128, 0), (0, 4096), (4096, 0)]
for rep in range(3):
    for (o, i) in cases:
        out, keys = out_all[o:o + n], keys_all[i:i + n]
        kw = dict(uncompressed_size=4, dictionary_index_size=1, index_size=4, stream=st)
        cfg = schemes.dictionary(out, keys, entries, n, 256, **kw)
        ms = sw.timeit(lambda: schemes.dictionary(out, keys, entries, n, 256, launch_config=cfg, **kw))
        print("rep %d  out +%-5d keys +%-5d elements  %8.3f ms %5.1f%%" % (rep, o, i, ms, 5 * n / ms / 1e6 / 80), flush=True)
