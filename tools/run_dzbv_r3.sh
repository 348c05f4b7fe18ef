# development aid (GPU box): DZB-V parity tests with the default library, then timings of library variants at several sizes:
#   bash tools/run_dzbv_r3.sh "28 31" default k1 k2 ...
SIZES="$1"; shift
python -m pytest tests/test_gpu_parity_rest.py tests/test_gpu_fuzz.py tests/test_gpu_large.py tests/test_gpu_unaligned.py tests/test_gpu_limits.py tests/test_gpu_sharding.py -m gpu -x -q -k "dzb or discard or unaligned or limits or sharding" 2>&1 | tail -5
for rep in 1 2; do
for v in "$@"; do
  if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  for s in $SIZES; do
    VERIFY=$([ $rep = 1 ] && echo 1 || echo 0) python tools/bench_one.py discard_zero_bytes_variable $s 20 2>&1 | grep -v amdgpu.ids | cut -c1-150
  done
done; done
