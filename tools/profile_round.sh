#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): rocprofv3 kernel-trace/stats of the default bench command, then
# separate --pmc passes (FETCH_SIZE, WRITE_SIZE — they do not fit one pass) over every workload and the
# calibration kernels.  Raw outputs go to gpurun_out/prof/; tools/summarize_profiles.py turns them into
# the committed summaries under profiles/.
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=gpurun_out/prof
if [ -z "${SKIP_PMC:-}" ]; then rm -rf $OUT; fi; mkdir -p $OUT; rm -rf $OUT/trace
hipcc --offload-arch=gfx950 -O3 tools/pmc_calibrate.hip -o /tmp/pmc_calibrate
# 1. kernel trace + stats of the bench command itself (ONLY_PMC=1 skips it)
if [ -z "${ONLY_PMC:-}" ]; then
timeout 1200 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sharded none --extras none --schemes coverage --details $OUT/bench_under_rocprof.details.json > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
# ... and the headline workload ALONE (the Zipf entry of the full line launches the same kernel: its dispatches would mix into "the last 20")
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_headline -o bench --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --sharded none --extras none --schemes none --details $OUT/bench_headline_under_rocprof.details.json > $OUT/bench_headline_under_rocprof.json 2> $OUT/bench_headline_under_rocprof.err
fi
# 2. PMC passes (separate runs); WORKLOADS="a b" overrides the list; SKIP_PMC=1 keeps the passes of an earlier call
if [ -n "${SKIP_PMC:-}" ]; then exit 0; fi
WORKLOADS=${WORKLOADS:-dictionary frame_of_reference delta_patched run_length_encoding run_position_encoding_relative run_position_encoding_absolute discard_zero_bytes_variable discard_zero_bytes_variable_3bit discard_zero_bytes_fixed model incidence_bitmaps delta_fused_patches dictionary_fused_patches frame_of_reference_fused_patches delta_fused_compressed_patches model_fused_patches discard_zero_bytes_fixed_fused_patches}
for counter in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $counter -d $OUT/pmc_$counter/calibrate -o cal --output-format csv -- /tmp/pmc_calibrate > $OUT/cal_$counter.log 2>&1
  for w in $WORKLOADS; do
    size=30; case $w in discard_zero_bytes_variable*|incidence_bitmaps) size=28;; esac
    # (only the decoders' dispatches are counted: with every kernel of the workload's generation under the counters a pass takes minutes)
    VERIFY=0 timeout 600 rocprofv3 --pmc $counter --kernel-include-regex "cuda::kernels" -d $OUT/pmc_$counter/$w -o pmc --output-format csv -- python3 tools/bench_one.py $w $size 3 > $OUT/pmc_${counter}_$w.log 2>&1
  done
done
find $OUT -name "*.csv" | head -50
