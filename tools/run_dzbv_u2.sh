# development aid (GPU box): DZB-variable into 2- and 4-byte elements for library variants
for rep in 1 2; do
for v in "$@"; do
  if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  python - "$v" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import sweep_all as sw
for I, U, CS, bits, mn in ((4, 2, 4, 2, 0), (8, 2, 4, 2, 0), (4, 2, 8, 2, 0), (4, 4, 4, 2, 1), (8, 8, 4, 3, 1)):
    ms, nbytes, ok = sw.run_dzb_variable(I, U, CS, 1 << 28, 1024, bits, mn)
    print("%-8s dzbv<%d,%d,%d> %d-bit min %d %8.3f ms %5.1f%% %s" % (sys.argv[1], I, U, CS, bits, mn, ms, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
PY
done; done
