# development aid (GPU box): RLE / RPE with 1- and 2-byte elements for library variants
for rep in 1 2; do
for v in "$@"; do
  if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  python - "$v" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import sweep_all as sw
n = 1 << 30
for name, fn in (("rle<4,1,1>", lambda: sw.run_rle(4, 1, 1, 4, n)), ("rle<4,1,2>", lambda: sw.run_rle(4, 1, 2, 4, n)), ("rpe<4,1,2,rel>", lambda: sw.run_rpe(4, 1, 2, True, 4, n)),
                 ("rle<8,1,8>", lambda: sw.run_rle(8, 1, 8, 8, n)), ("rle<4,2,2>", lambda: sw.run_rle(4, 2, 2, 4, n)), ("rpe<4,2,2,rel>", lambda: sw.run_rpe(4, 2, 2, True, 4, n)),
                 ("rle<4,8,2>", lambda: sw.run_rle(4, 8, 2, 4, n)), ("rle<4,4,2>", lambda: sw.run_rle(4, 4, 2, 4, n))):
    ms, nbytes, ok = fn()
    print("%-8s %-16s %8.3f ms %5.1f%% %s" % (sys.argv[1], name, ms, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
PY
done; done
