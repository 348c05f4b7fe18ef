# development aid (GPU box): RLE / RPE parity tests, then the run-length sweep of tools/param_sweep.py
python -m pytest tests/test_gpu_parity_rest.py tests/test_gpu_fuzz.py tests/test_gpu_large.py tests/test_gpu_unaligned.py tests/test_gpu_limits.py tests/test_gpu_sharding.py tests/test_gpu_baseline_configs.py tests/test_golden.py -m gpu -x -q -k "rle or rpe or runs or unaligned or limits or sharding or golden or run_positions" 2>&1 | tail -4
python - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import sweep_all as sw
n = 1 << 28
def show(name, res):
    ms, nbytes, ok = res
    print("%-50s %8.3f ms %8.1f GB/s %5.1f%% %s" % (name, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, "ok" if ok else "MISMATCH"), flush=True)
for mean_run in (2, 3, 4, 6, 8, 12, 16, 32, 128):
    show("rle<4,4,2> mean_run=%d L=4096" % mean_run, sw.run_rle(4, 4, 2, 4, n, 4096, mean_run))
for mean_run in (2, 4, 8, 32):
    show("rpe<4,4,2,rel> mean_run=%d L=4096" % mean_run, sw.run_rpe(4, 4, 2, True, 4, n, 4096, mean_run))
    show("rpe<4,4,4,abs> mean_run=%d L=4096" % mean_run, sw.run_rpe(4, 4, 4, False, 4, n, 4096, mean_run))
for mean_run in (2, 4, 32):
    show("rle<4,1,1> mean_run=%d" % mean_run, sw.run_rle(4, 1, 1, 4, n, 4096, mean_run))
    show("rle<4,2,2> mean_run=%d" % mean_run, sw.run_rle(4, 2, 2, 4, n, 4096, mean_run))
    show("rle<4,8,2> mean_run=%d" % mean_run, sw.run_rle(4, 8, 2, 4, n, 4096, mean_run))
    show("rle<4,4,4> mean_run=%d" % mean_run, sw.run_rle(4, 4, 4, 4, n, 4096, mean_run))
for mean_run in (4, 32):
    show("rle<4,4,2> mean_run=%d L=65536" % mean_run, sw.run_rle(4, 4, 2, 4, n, 65536, mean_run))
    show("rle<4,4,2> mean_run=%d L=1024" % mean_run, sw.run_rle(4, 4, 2, 4, n, 1024, mean_run))
PY
