"""GPU box: the decoders at the EDGES of their parameters — tiny and odd segment lengths, segment lengths that defeat the 16-byte paths, dictionaries of 1 ... 2^20 entries,
1 ... 256 bitmaps, mean runs of 1 ... 10^5 — one line each, 2^27 elements unless said: where does a decoder fall off a cliff?   python tools/holes.py [name ...]"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import sweep_all as sw

N = 1 << int(os.environ.get("LOG2N", "27"))
only = sys.argv[1:]
def line(name, what, f):
    if only and not any(o in name for o in only):
        return
    try:
        ms, nbytes, ok = f()
        print("%-34s %-28s %8.3f ms %5.1f%% %s" % (name, what, ms, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
    except Exception as e:
        print("%-34s %-28s refused: %s" % (name, what, str(e).split("\n")[0][:90]), flush=True)

for L in (2, 6, 10, 100, 1001 * 2, 4094):   # not multiples of the lanes' groups: the element-wise segmented scan
    line("delta<4,4,1>", "L=%d" % L, lambda: sw.run_delta(4, 4, 1, N // L * L, L))
    line("delta<4,8,2>", "L=%d" % L, lambda: sw.run_delta(4, 8, 2, N // L * L, L))
for L in (4, 8, 12, 20, 36, 100):
    line("frame_of_reference<4,4,1,linear>", "L=%d" % L, lambda: sw.run_for(4, 4, 1, "linear", 4, N // L * L, L))
    line("frame_of_reference<4,4,2,const>", "L=%d" % L, lambda: sw.run_for(4, 4, 2, "constant", 4, N // L * L, L))
for (D, entries) in ((1, 1), (1, 2), (1, 16), (1, 256), (2, 257), (2, 4096), (2, 65536), (4, 65537), (4, 1 << 20), (4, 1 << 24)):
    line("dictionary<4,4,%d>" % D, "%d entries" % entries, lambda: sw.run_dictionary(4, 4, D, N, entries_n=entries))
    line("dictionary<4,8,%d>" % D, "%d entries" % entries, lambda: sw.run_dictionary(4, 8, D, N, entries_n=entries))
for nb in (1, 2, 3, 5, 16, 17, 64, 65, 200, 256):
    line("incidence_bitmaps<4,4,2,dict>", "%d bitmaps" % nb, lambda: sw.run_ib(4, 4, 2, True, N, nb))
for mean in (1, 2, 3, 5, 8, 100, 1000, 100000):
    line("run_length_encoding<4,4,2,4>", "mean run %d" % mean, lambda: sw.run_rle(4, 4, 2, 4, N, 4096, mean))
    line("run_position_encoding<4,4,4,abs>", "mean run %d" % mean, lambda: sw.run_rpe(4, 4, 4, False, 4, N, 4096, mean))
for L in (3, 10, 30, 100, 250, 1000, 1023):   # not multiples of the lanes' groups, or of the size containers' fields
    if L % 16 == 0: continue
    line("discard_zero_bytes_variable<4,4,4>", "L=%d" % L, lambda: sw.run_dzb_variable(4, 4, 4, N // L * L // 16 * 16 if False else N // (L * 16) * (L * 16), L))
    line("discard_zero_bytes_variable<4,8,4>", "L=%d bits=3" % L, lambda: sw.run_dzb_variable(4, 8, 4, N // (L * 10) * (L * 10), L, bits=3))
for (bits, min_size) in ((1, 1), (1, 3), (2, 0), (4, 0), (4, 1)):
    line("discard_zero_bytes_variable<4,8,4>", "L=1024 bits=%d min=%d" % (bits, min_size), lambda: sw.run_dzb_variable(4, 8, 4, N, 1024, bits=bits, min_size=min_size))
