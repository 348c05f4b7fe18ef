# fused Model, all instantiations: a seek probing 64 positions a page apart (variants/wide.so: -DGIDDY_MODEL_PATCHED_WIDE_GUESS=1) against 16 four pages apart (variants/before.so), 2^30 elements
for f in 0.001 0.005 0.02; do
  for v in before wide before wide; do
    export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so
    PATCH_FRACTION=$f SWEEP_REPS=9 timeout 600 python tools/sweep_one.py "model::patched" 30 2>&1 | grep "kernel_t" | sed -E "s/^/$v patches $f: /; s/ +/ /g; s/decompression:://; s/cuda::functors::unary::parametric_model:://"
  done
done
