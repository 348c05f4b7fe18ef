# development aid (GPU box): the C4 bench workloads for library variants:  bash tools/run_c4_variants.sh default all ...
for rep in 1 2 3; do
for v in "$@"; do
  if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  for w in run_length_encoding run_position_encoding_relative run_position_encoding_absolute; do
    echo "$v $w $(VERIFY=0 python tools/bench_one.py $w 30 20 2>/dev/null | tail -1)"
  done
done; done
