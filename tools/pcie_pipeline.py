"""PCIe-inclusive rate of the Dictionary path: compressed keys start in pinned host memory.
    python tools/pcie_pipeline.py [log2n=30]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from libgiddy_amd import pipeline

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << log2n
g = torch.Generator(); g.manual_seed(2)
keys_host = torch.randint(0, 256, (n,), dtype=torch.uint8, generator=g).pin_memory()
entries = (torch.randperm(1 << 24, device="cuda")[:256] * 251 + 7).to(torch.int32)
out = torch.empty(n, dtype=torch.int32, device="cuda")
# plain H2D ceiling
dev = torch.empty(n, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize(); t = time.perf_counter(); dev.copy_(keys_host, non_blocking=True); torch.cuda.synchronize()
h2d = time.perf_counter() - t
print("H2D copy of %d MiB of keys: %.1f ms = %.1f GB/s" % (n >> 20, h2d * 1e3, n / h2d / 1e9))
for chunk_log2, depth in ((24, 2), (24, 3), (26, 3), (28, 2)):
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for s in pipeline.dictionary_from_host(keys_host, entries, out, chunk_elements=1 << chunk_log2, depth=depth):
            s.synchronize()
        best = min(best, time.perf_counter() - t)
    ok = bool((out[: 1 << 24] == entries[keys_host[: 1 << 24].cuda().long()]).all()) and bool((out[-(1 << 20):] == entries[keys_host[-(1 << 20):].cuda().long()]).all())
    print("pipelined host->device decode, chunks of 2^%d, %d streams: %.1f ms -> %.1f GB/s decompressed (%.1f GB/s of keys over PCIe) ok=%s"
          % (chunk_log2, depth, best * 1e3, 4 * n / best / 1e9, n / best / 1e9, ok))
