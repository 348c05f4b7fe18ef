# the fused kernels with 2-byte (and 1-byte) outputs — rows of 512 (1024) elements: where do their two builds cross?  (variants many_always / many_never; 2^28 elements)
PATTERN=${PATTERN:-'model::patched::kernel_t<4u, short|dictionary::patched::kernel_t<4u, 2u, 1u>|dictionary::patched::kernel_t<4u, 1u, 1u>|fixed_width::patched::kernel_t<4u, 2u, 1u|delta::patched::kernel_t<4u, 2u, 1u>|frame_of_reference::patched::kernel_t<4u, short, signed char, .*linear|dictionary::patched::kernel_t<4u, 8u, 1u>|fixed_width::patched::kernel_t<4u, 8u, 1u'}
for f in ${FRACTIONS:-0.001 0.0025 0.005 0.01 0.02}; do
  for v in many_always many_never; do
    export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so
    PATCH_FRACTION=$f SWEEP_REPS=7 timeout 600 python tools/sweep_one.py "$PATTERN" 28 2>&1 | grep "kernel_t" | sed -E "s/^/$v patches $f: /; s/ +/ /g; s/decompression:://; s/cuda::functors::unary::parametric_model:://; s/\(cuda::endianness_t\)1, \(unsigned short\)32//"
  done
done
