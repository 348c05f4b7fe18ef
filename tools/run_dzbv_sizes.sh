# development aid (GPU box): DZB-V <8,8,4> over column sizes for library variants:  bash tools/run_dzbv_sizes.sh "28 29 30 31" default r2 ...
# (a variant named NAME:static runs variants/NAME.so with GIDDY_ORDERED_TILES=0)
SIZES="$1"; shift
for rep in 1 2; do
for spec in "$@"; do
  v=${spec%%:*}
  if [ "$spec" != "$v" ]; then export GIDDY_ORDERED_TILES=0; else unset GIDDY_ORDERED_TILES; fi
  if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  for s in $SIZES; do
    VERIFY=$([ $rep = 1 ] && echo 1 || echo 0) timeout 300 python tools/bench_one.py discard_zero_bytes_variable $s 10 2>&1 | grep -v amdgpu.ids | sed "s/discard_zero_bytes_variable/dzbv 2^$s/;s/^default /$spec /" | cut -c1-140
  done
done; done
