PATTERN='frame_of_reference::patched::kernel_t<4u, int, short, .*linear'
for f in 0.005 0.02 0.05 0.2; do
  for v in default for_r4 for_r2 for_stage8; do
    if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
    PATCH_FRACTION=$f SWEEP_REPS=7 timeout 400 python tools/sweep_one.py "$PATTERN" 28 2>&1 | grep "kernel_t" | sed -E "s/^/$v patches $f: /; s/ +/ /g; s/decompression:://; s/cuda::functors::unary::parametric_model:://"
  done
done
