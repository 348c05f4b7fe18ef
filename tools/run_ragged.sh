# development aid (GPU box): RLE / RPE at segment lengths that are not multiples of 128 bytes, for library variants
for v in "$@"; do
  if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  python - "$v" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import sweep_all as sw
n = 1 << 28
for L in (1000, 3000, 3500, 4000, 4096, 5000, 10000):
    m = n // L * L
    for name, fn in (("rle<4,4,2>", lambda: sw.run_rle(4, 4, 2, 4, m, L)), ("rpe<4,4,2,rel>", lambda: sw.run_rpe(4, 4, 2, True, 4, m, L))):
        ms, nbytes, ok = fn()
        print("%-8s %-16s L=%-6d %8.3f ms %5.1f%% %s" % (sys.argv[1], name, L, ms, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
PY
done
