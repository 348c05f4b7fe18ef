# fused fixed-width discard-zero-bytes, the instantiations whose residency is throttled (8-byte outputs: 2 workgroups per CU, <4,2> and <2,1>: 4): throttled whatever the patches
# (a build of the wrapper with the rule taken out: the macro that switched it is gone, the measurement is in profiles/r05_patch_density.txt) against not throttled where the ManyPatches build is launched (the library), 2^28 elements
for f in 0.005 0.02 0.05 0.2; do
  PATCH_FRACTION=$f SWEEP_REPS=7 timeout 600 python tools/sweep_one.py "dictionary::patched::kernel_t<4u, 8u, [12]u>" 28 2>&1 | grep "kernel_t" | sed -E "s/^/default patches $f: /; s/ +/ /g; s/decompression:://"
  for v in throttled default; do
    if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
    PATCH_FRACTION=$f SWEEP_REPS=7 timeout 600 python tools/sweep_one.py "fixed_width::patched::kernel_t<4u, (8u, 1u|8u, 4u|4u, 2u|2u, 1u)" 28 2>&1 | grep "kernel_t" | sed -E "s/^/$v patches $f: /; s/ +/ /g; s/decompression:://; s/\(cuda::endianness_t\)1, \(unsigned short\)32//"
  done
done
