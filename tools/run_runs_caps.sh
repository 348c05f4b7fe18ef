# development aid (GPU box): RLE/RPE with 4-byte descriptors under different residency caps
for rep in 1 2; do
for v in "$@"; do
  if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  python - "$v" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import sweep_all as sw
for name, fn in (("rpe<4,4,4,abs>", lambda: sw.run_rpe(4, 4, 4, False, 4, 1 << 30)), ("rle<4,4,4>", lambda: sw.run_rle(4, 4, 4, 4, 1 << 30)), ("rpe<4,8,4,abs>", lambda: sw.run_rpe(4, 8, 4, False, 4, 1 << 30))):
    ms, nbytes, ok = fn()
    print("%-8s %-16s %8.3f ms %5.1f%% %s" % (sys.argv[1], name, ms, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
PY
done; done
