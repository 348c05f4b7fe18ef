# GPU box: the fused decompress-and-patch kernels over patch densities, for library variants:   bash tools/patch_density.sh default one_at_a_time ...
# (PATCH_FRACTION: tools/sweep_all.py make_patches; 2^28 elements)
PATTERN='delta::patched::kernel_t<4u, 8u, 2u>|dictionary::patched::kernel_t<4u, 4u, 1u>|frame_of_reference::patched::kernel_t<4u, int, short, .*linear|model::patched::kernel_t<4u, int, .*linear|fixed_width::patched::kernel_t<4u, 4u, 1u'
for f in ${FRACTIONS:-0.001 0.01 0.05 0.2 0.5}; do
  for v in "$@"; do
    if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
    PATCH_FRACTION=$f SWEEP_REPS=${SWEEP_REPS:-9} timeout 400 python tools/sweep_one.py "$PATTERN" ${LOG2N:-28} 2>&1 | grep "kernel_t" | sed -E "s/^/$v patches $f: /; s/ +/ /g; s/decompression:://; s/cuda::functors::unary::parametric_model:://; s/\(util::endianness_t\)1, \(unsigned short\)32//"
  done
done
