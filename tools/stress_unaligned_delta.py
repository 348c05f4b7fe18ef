"""GPU box: random shapes against Delta at odd element offsets with segments LONGER than a window (decompress_windows_at_any_alignment: the pre-pass's tail sums and
bases, the register-only windows), checked against a device-side restatement (segment-wise cumulative sums).    python tools/stress_unaligned_delta.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
import sweep_all as sw
from libgiddy_amd import schemes

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rnd = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
st = torch.cuda.current_stream().cuda_stream
started, cases, failures = time.time(), 0, 0
while time.time() - started < budget:
    U, C = [(2, 1), (4, 1), (4, 2), (8, 1), (8, 2), (8, 4)][rnd.integers(0, 6)]
    window = 32 * 64 * 16 // U
    n = int(rnd.integers(1, 1 << int(rnd.integers(10, 25))))
    kind = rnd.integers(0, 4)
    L = int(window * rnd.integers(1, 9) + rnd.integers(0, 3) * (16 // U)) if kind == 0 else int(rnd.integers(window + 1, max(window + 2, 4 * n))) if kind == 1 \
        else int(rnd.integers(window + 1, 40 * window)) if kind == 2 else n + int(rnd.integers(0, 5))
    L = max(L, window + 1)
    out_offset, in_offset = int(rnd.integers(0, 40)), int(rnd.integers(0, 40))
    if out_offset % (16 // U) == 0 and in_offset % (16 // U) == 0:
        out_offset += 1
    g = sw.gen(cases)
    d_all = sw.rand_bits(n + in_offset, C, g)
    d = d_all[in_offset:]
    du = sw.unsigned(d, C)
    segments = -(-n // L)
    anchors64 = torch.randint(0, 1 << 40, (segments,), generator=g, device=sw.DEV, dtype=torch.int64)
    anchors = anchors64.to(sw.INT[U])
    out_all = torch.full((n + out_offset + 8,), -1 if U < 8 else -1, dtype=sw.INT[U], device=sw.DEV)
    out = out_all[out_offset: out_offset + n]
    schemes.delta(out, d, anchors, n, L, uncompressed_size=U, compressed_size=C, index_size=4, stream=st)
    padded = torch.zeros(segments * L, dtype=torch.int64, device=sw.DEV)
    padded[:n] = du
    expect = (torch.cumsum(padded.view(segments, L), dim=1) + anchors64[:, None]).view(-1)[:n].to(sw.INT[U]) if segments * L < (1 << 28) else None
    if expect is None:
        continue
    ok = bool((out == expect).all()) and bool((out_all[:out_offset] == -1).all()) and bool((out_all[out_offset + n:] == -1).all())
    cases += 1
    if not ok:
        failures += 1
        print("FAILED: U=%d C=%d n=%d L=%d out+%d in+%d" % (U, C, n, L, out_offset, in_offset), flush=True)
print("%d random unaligned long-segment Delta cases, %d failures" % (cases, failures))
sys.exit(1 if failures else 0)
