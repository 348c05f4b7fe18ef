# development aid (GPU box): the fixed-rate decoders' bench workloads for library variants:  bash tools/run_stream_variants.sh default po4 ...
for rep in 1 2 3; do
for v in "$@"; do
  if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  for w in ${WORKLOADS:-dictionary frame_of_reference discard_zero_bytes_fixed model}; do
    echo "$v $(VERIFY=0 python tools/bench_one.py $w 30 20 2>/dev/null | tail -1 | cut -c36-150)"
  done
done; done
