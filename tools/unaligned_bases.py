"""GPU box: what a column handed over at an odd ELEMENT offset costs (output and/or input one element past a 16-byte boundary): the decoders' element-wise fall-backs.
    python tools/unaligned_bases.py [log2 n]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
import sweep_all as sw
from libgiddy_amd import schemes

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 27)
st = torch.cuda.current_stream().cuda_stream
def report(name, how, ms, nbytes, ok):
    print("%-28s %-34s %8.3f ms %5.1f%% %s" % (name, how, ms, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)

g = sw.gen(1)
# Dictionary <4,4,1>
entries = sw.rand_bits(256, 4, g); keys_all = sw.rand_bits(n + 2048, 1, g, high=256); out_all = torch.empty(n + 2048, dtype=torch.int32, device=sw.DEV)
for (o, i) in ((0, 0), (1, 0), (0, 1), (1, 1), (4, 4), (32, 32), (1024, 1024), (1025, 3)):
    out, keys = out_all[o:o + n], keys_all[i:i + n]
    kw = dict(uncompressed_size=4, dictionary_index_size=1, index_size=4, stream=st)
    cfg = schemes.dictionary(out, keys, entries, n, 256, **kw)
    ms = sw.timeit(lambda: schemes.dictionary(out, keys, entries, n, 256, launch_config=cfg, **kw))
    report("dictionary<4,4,1>", "out +%d, keys +%d elements" % (o, i), ms, 5 * n, bool((entries[sw.unsigned(keys, 1)] == out).all()))
# Frame of reference <4,4,2,constant>, segments of 1024
Lf = 1024
offsets_all = sw.rand_bits(n + 2048, 2, g); refs = sw.rand_bits(n // Lf + 4, 4, g, high=1 << 20)
for (o, i) in ((0, 0), (1, 0), (1, 1), (4, 4), (1025, 3)):
    out, offs = out_all[o:o + n], offsets_all[i:i + n]
    kw = dict(uncompressed_size=4, compressed_size=2, model="constant", index_size=4, stream=st)
    cfg = schemes.frame_of_reference(out, offs, refs, n, Lf, **kw)
    ms = sw.timeit(lambda: schemes.frame_of_reference(out, offs, refs, n, Lf, launch_config=cfg, **kw))
    expect = (refs[:n // Lf].view(-1, 1) + offs.view(-1, Lf).to(torch.int32)).view(-1)
    report("frame_of_reference<4,4,2,c>", "out +%d, offsets +%d elements" % (o, i), ms, 6 * n + 4 * (n // Lf), bool((expect == out).all()))
# Delta, segments of up to 32 rows (the any-alignment windows) and beyond (round 6: windows whose base comes from a pre-pass)
d_all = {1: sw.rand_bits(n + 64, 1, g), 2: sw.rand_bits(n + 64, 2, g)}
outs = {4: out_all, 8: torch.empty(n + 2048, dtype=torch.int64, device=sw.DEV)}
for (U, C, L) in ((4, 1, 4096), (4, 1, 16384), (4, 1, 65536), (8, 2, 4096), (8, 2, 8192), (8, 2, 65536), (8, 2, 10000), (4, 1, 1 << 20)):
    m = n // L * L
    for (o, i) in ((0, 0), (1, 0), (0, 1), (1, 1)):
        d = d_all[C][i:i + m]; out = outs[U][o:o + m]
        du = sw.unsigned(d, C); per_seg = du.view(m // L, L).sum(dim=1)
        anchors64 = (1 << 20) + torch.cumsum(per_seg, 0) - per_seg; anchors = anchors64.to(sw.INT[U])
        kw = dict(uncompressed_size=U, compressed_size=C, index_size=4, stream=st)
        cfg = schemes.delta(out, d, anchors, m, L, **kw)
        ms = sw.timeit(lambda: schemes.delta(out, d, anchors, m, L, launch_config=cfg, **kw))
        ok = True
        for lo in range(0, m, max(L, (1 << 24) // L * L)):
            hi = min(m, lo + max(L, (1 << 24) // L * L))
            expect = (torch.cumsum(du[lo:hi].view(-1, L), dim=1) + anchors64[lo // L: hi // L, None]).view(-1).to(sw.INT[U])
            ok = ok and bool((expect == out[lo:hi]).all())
        report("delta<4,%d,%d> L=%d" % (U, C, L), "out +%d, deltas +%d elements" % (o, i), ms, (U + C) * m + U * (m // L), ok)
# run lengths <4,4,2,4>
lengths, starts = sw._runs(n, 32, g, 65535); count = len(lengths)
values = (torch.arange(count, device=sw.DEV) * 2654435761 % (1 << 31)).to(torch.int32)
seg = torch.arange(0, n, L, device=sw.DEV); anchor = torch.searchsorted(starts, seg, right=True) - 1
for o in (0, 1):
    out = out_all[o:o + n]
    args = (values, lengths.to(torch.int16), anchor.to(torch.int32), (seg - starts[anchor]).to(torch.int16), L, len(seg), count, n)
    kw = dict(uncompressed_size=4, run_length_size=2, index_size=4, run_index_size=4, stream=st)
    cfg = schemes.run_length_encoding(out, *args, **kw)
    ms = sw.timeit(lambda: schemes.run_length_encoding(out, *args, launch_config=cfg, **kw))
    ok = bool((values[torch.searchsorted(starts, torch.arange(0, min(n, 1 << 24), device=sw.DEV), right=True) - 1] == out[:1 << 24]).all())
    report("run_length_encoding<4,4,2,4>", "out +%d elements" % o, ms, 4 * n + 6 * count + 6 * len(seg), ok)

# run positions <4,4,4,absolute>
positions = starts.to(torch.int32)
for o in (0, 1):
    out = out_all[o:o + n]
    args = (values, positions, anchor.to(torch.int32), L, len(seg), count, n)
    kw = dict(uncompressed_size=4, position_size=4, relative=False, index_size=4, run_index_size=4, stream=st)
    try:
        cfg = schemes.run_position_encoding(out, *args, **kw)
        ms = sw.timeit(lambda: schemes.run_position_encoding(out, *args, launch_config=cfg, **kw))
        ok = bool((values[torch.searchsorted(starts, torch.arange(0, min(n, 1 << 24), device=sw.DEV), right=True) - 1] == out[:1 << 24]).all())
        report("run_position_encoding<4,4,4,abs>", "out +%d elements" % o, ms, 4 * n + 8 * count + 4 * len(seg), ok)
    except Exception as e:
        print("run_position_encoding: ", str(e)[:200])
