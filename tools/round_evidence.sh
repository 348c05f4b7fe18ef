#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round's evidence with the library as committed — the GPU test suite, the driver's bench command, the coverage bench,
# the profile set (tools/profile_round.sh), the fused wrappers' regret, the unaligned columns, and the sweep of every instantiation.  Raw outputs under
# gpurun_out/$NAME/; tools/summarize_profiles.py and a few cp turn them into profiles/<round>_*.     NAME=r6final bash tools/round_evidence.sh
NAME=${NAME:-evidence}
OUT=gpurun_out/$NAME; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1; echo tests rc=$?; tail -2 $OUT/gputests.log
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.stdout 2> $OUT/bench_default.stderr; echo bench rc=$?
cp bench_details.json $OUT/bench_default_details.json; tail -c 2500 $OUT/bench_default.stdout
python3 bench.py --schemes coverage --sharded none --details $OUT/bench_coverage_details.json > $OUT/bench_coverage.stdout 2> $OUT/bench_coverage.stderr; echo coverage rc=$?
bash tools/profile_round.sh > $OUT/profile_round.log 2>&1; echo profile rc=$?
python tools/fused_regret.py 28 > $OUT/fused_regret.txt 2>&1; tail -1 $OUT/fused_regret.txt
python tools/unaligned_bases.py 27 > $OUT/unaligned_bases.txt 2>&1
python tools/sweep_all.py 30 > $OUT/sweep_all.txt 2>&1; grep -c " ok" $OUT/sweep_all.txt; grep -c MISMATCH $OUT/sweep_all.txt
