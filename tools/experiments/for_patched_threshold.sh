# fused FOR: its ManyPatches build (rows in flight: tuning.hpp) forced against never taken, around where the two cross
PATTERN='frame_of_reference::patched::kernel_t<4u, int, short, .*linear'
for f in 0.005 0.0075 0.01 0.0125 0.015 0.02; do
  for v in many_always many_never; do
    export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so
    PATCH_FRACTION=$f SWEEP_REPS=7 timeout 400 python tools/sweep_one.py "$PATTERN" 28 2>&1 | grep "kernel_t" | sed -E "s/^/$v patches $f: /; s/ +/ /g; s/decompression:://; s/cuda::functors::unary::parametric_model:://"
  done
done
