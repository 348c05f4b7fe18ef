#!/bin/bash
# Runs ON THE GPU BOX: average latency of vector-memory, LDS and scalar-memory instructions of one workload's decoder
# (SQ_INST_LEVEL_* accumulate the instructions in flight every cycle; divided by the instruction count = mean cycles in flight).
#   tools/pmc_latency.sh WORKLOAD LOG2N KERNEL_SUBSTRING      (environment, e.g. MEAN_RUN=4, is passed through)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
W=$1; N=$2; K=$3
OUT=gpurun_out/pmc_latency/$W${MEAN_RUN:+_m$MEAN_RUN}; rm -rf $OUT; mkdir -p $OUT
i=0
for P in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAVES SQ_WAVE_CYCLES" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES" "SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_ANY"; do
  i=$((i+1))
  VERIFY=0 rocprofv3 --pmc $P -d $OUT/p$i -o pmc --output-format csv -- python3 tools/bench_one.py $W $N 3 > $OUT/p$i.log 2>&1
done
python3 - "$OUT" "$K" <<'PY'
import csv, glob, os, sys
out, kernel = sys.argv[1], sys.argv[2]
t = {}
for path in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    s, c = {}, {}
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            s[r["Counter_Name"]] = s.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]); c[r["Counter_Name"]] = c.get(r["Counter_Name"], 0) + 1
    for k in s: t[k] = s[k] / c[k]
for k in sorted(t): print("%-24s %16.0f" % (k, t[k]))
for unit in ("VMEM", "LDS", "SMEM"):
    if t.get("SQ_INSTS_" + unit): print("mean cycles in flight per %s instruction: %.0f" % (unit, t["SQ_INST_LEVEL_" + unit] / t["SQ_INSTS_" + unit]))
PY
