# development aid (GPU box): run-length sweep for library variants:  bash tools/run_runs_variants.sh default wd ...
for v in "$@"; do
  if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  python - "$v" <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import sweep_all as sw
n = 1 << 28
tag = sys.argv[1]
def show(name, res):
    ms, nbytes, ok = res
    print("%-8s %-44s %8.3f ms %8.1f GB/s %5.1f%% %s" % (tag, name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
for mean_run in (2, 4, 6, 8, 16, 32):
    show("rle<4,4,2> mean_run=%d" % mean_run, sw.run_rle(4, 4, 2, 4, n, 4096, mean_run))
for mean_run in (4, 32):
    show("rpe<4,4,2,rel> mean_run=%d" % mean_run, sw.run_rpe(4, 4, 2, True, 4, n, 4096, mean_run))
    show("rpe<4,4,4,abs> mean_run=%d" % mean_run, sw.run_rpe(4, 4, 4, False, 4, n, 4096, mean_run))
    show("rle<4,8,2> mean_run=%d" % mean_run, sw.run_rle(4, 8, 2, 4, n, 4096, mean_run))
    show("rle<4,2,2> mean_run=%d" % mean_run, sw.run_rle(4, 2, 2, 4, n, 4096, mean_run))
PY
done
