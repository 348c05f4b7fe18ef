# fused Model: resident workgroups per CU (GIDDY_MODEL_PATCHED_RESIDENT_BLOCKS_CAP: 3 as built | 5 | 8) over the patch density
PATTERN='model::patched::kernel_t<4u, int, .*linear'
for f in 0.001 0.01 0.03 0.1 0.2 0.5; do
  for v in default model_cap5 model_cap8; do
    if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
    PATCH_FRACTION=$f SWEEP_REPS=7 timeout 400 python tools/sweep_one.py "$PATTERN" 28 2>&1 | grep "kernel_t" | sed -E "s/^/$v patches $f: /; s/ +/ /g; s/decompression:://; s/cuda::functors::unary::parametric_model:://"
  done
done
