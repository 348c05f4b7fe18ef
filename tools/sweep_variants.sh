# development aid (GPU box): tools/sweep_variants.sh PATTERN LOG2N variant [variant ...] — sweep the instantiations matching PATTERN with each library variant
PATTERN=$1; N=$2; shift 2
for v in "$@"; do
  if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  bash tools/sweep_some.sh "$PATTERN" $N 2>&1 | grep kernel_t | sed "s/^/$v /"
done
