"""Quick per-kernel timing on one GPU (development aid; bench.py is the contract benchmark)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from libgiddy_amd import schemes

def timeit(fn, reps=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in evs)
    return ts[len(ts)//2] * 1e-3

def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    n = 1 << logn
    st = torch.cuda.current_stream().cuda_stream
    dev = "cuda"
    # copy ceiling
    a = torch.empty(n, dtype=torch.int32, device=dev); b = torch.empty_like(a)
    t = timeit(lambda: b.copy_(a))
    print("copy d2d int32: %.1f GB/s (read+write)" % (2 * 4 * n / t / 1e9))
    del a, b
    # dictionary u8 -> u32
    keys = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev)
    d = torch.randint(0, 2**31 - 1, (256,), dtype=torch.int32, device=dev)
    out = torch.empty(n, dtype=torch.int32, device=dev)
    cfg = schemes.dictionary(out, keys, d, n, 256, stream=st)
    t = timeit(lambda: schemes.dictionary(out, keys, d, n, 256, stream=st, launch_config=cfg))
    ok = bool((out == d[keys.long()]).all()) if logn <= 30 else None
    print("dictionary <4,4,1>: %.3f ms  %.1f GB/s algorithmic (5 B/elem), %.1f GB/s decompressed  ok=%s cfg=%s" % (t*1e3, 5*n/t/1e9, 4*n/t/1e9, ok, cfg))
    t = timeit(lambda: schemes.dictionary(out, keys, d, n, 256, stream=st, cache_dictionary_in_shared_mem=False))
    print("dictionary <4,4,1> no LDS: %.3f ms  %.1f GB/s algorithmic" % (t*1e3, 5*n/t/1e9))
    del keys, out
    # FOR i16 -> i32, constant
    comp = torch.randint(-32768, 32767, (n,), dtype=torch.int16, device=dev)
    L = 4096
    coefs = torch.randint(-10**6, 10**6, (n // L, 1), dtype=torch.int32, device=dev)
    out = torch.empty(n, dtype=torch.int32, device=dev)
    cfg = schemes.frame_of_reference(out, comp, coefs, n, L, stream=st)
    t = timeit(lambda: schemes.frame_of_reference(out, comp, coefs, n, L, stream=st, launch_config=cfg))
    ok = bool((out == (comp.int() + coefs.repeat_interleave(L, 0).view(-1))).all()) if logn <= 30 else None
    print("FOR <4,i32,i16,const>: %.3f ms  %.1f GB/s algorithmic (6 B/elem) ok=%s cfg=%s" % (t*1e3, 6*n/t/1e9, ok, cfg))
    del comp, out, coefs
    # delta u16 -> u64
    dl = torch.randint(0, 65535, (n,), dtype=torch.int32, device=dev).to(torch.int16)  # bit pattern of u16
    pref = torch.cumsum((dl.int() & 0xFFFF).long(), 0) + (1 << 40)
    anchors = torch.cat([torch.tensor([1 << 40], device=dev, dtype=torch.int64), pref[L-1:-1:L]]).contiguous()
    out = torch.empty(n, dtype=torch.int64, device=dev)
    cfg = schemes.delta(out, dl, anchors, n, L, stream=st)
    t = timeit(lambda: schemes.delta(out, dl, anchors, n, L, stream=st, launch_config=cfg))
    ok = bool((out == pref).all())
    print("delta <4,8,2> L=4096: %.3f ms  %.1f GB/s algorithmic (10 B/elem) ok=%s cfg=%s" % (t*1e3, 10*n/t/1e9, ok, cfg))

main()
