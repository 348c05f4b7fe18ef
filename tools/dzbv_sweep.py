"""DZB-variable across instantiations, field widths and segment lengths (development aid; the library is chosen by GIDDY_LIBRARY_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import sweep_all as sw
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
tag = os.path.basename(os.environ.get("GIDDY_LIBRARY_PATH", "default"))
def show(name, res):
    ms, nbytes, ok = res
    print("%-10s %-44s %8.3f ms %8.1f GB/s %5.1f%% %s" % (tag, name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
for I, U, CS in [(4, 2, 4), (4, 4, 4), (4, 8, 4), (8, 2, 4), (8, 4, 4), (8, 8, 4), (4, 2, 8), (4, 4, 8), (4, 8, 8)]:
    show("dzbv<%d,%d,%d> 2-bit min 1 L=1024" % (I, U, CS), sw.run_dzb_variable(I, U, CS, n, 1024, 2, 1 if U > 2 else 0))
for bits, mn in [(1, 1), (2, 0), (3, 1), (4, 0)]:
    for L in (1024, 4096):
        show("dzbv<8,8,4> %d-bit min %d L=%d" % (bits, mn, L), sw.run_dzb_variable(8, 8, 4, n, L, bits, mn))
for L in (128, 256, 512, 2048, 4096, 65536, 1000, 3000):
    m = n // (L * 16) * (L * 16)
    show("dzbv<8,8,4> 2-bit min 1 L=%d" % L, sw.run_dzb_variable(8, 8, 4, m, L))
show("dzbv<4,4,4> 4-bit min 0 L=1024", sw.run_dzb_variable(4, 4, 4, n, 1024, 4, 0))
show("dzbv<4,4,8> 2-bit min 1 L=4096", sw.run_dzb_variable(4, 4, 8, n, 4096, 2, 1))
