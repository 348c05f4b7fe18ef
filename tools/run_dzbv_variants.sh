# development aid (GPU box): DZB-V parity tests, then timings of library variants:  bash tools/run_dzbv_variants.sh default old a2 ...
python -m pytest tests/test_gpu_parity_rest.py tests/test_gpu_fuzz.py tests/test_gpu_large.py tests/test_gpu_unaligned.py tests/test_gpu_limits.py tests/test_gpu_sharding.py -m gpu -x -q -k "dzb or unaligned or limits or sharding" 2>&1 | tail -3
for rep in 1 2; do
for v in "$@"; do
  if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  python tools/bench_one.py discard_zero_bytes_variable 28 20 2>&1 | cut -c1-150
  python tools/bench_one.py discard_zero_bytes_variable_3bit 28 20 2>&1 | cut -c1-150
done; done
