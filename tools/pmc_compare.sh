#!/bin/bash
# Runs ON THE GPU BOX: memory-side and SQ counters of the Delta decoder for a list of segment lengths, side by side — what differs between a fast and a slow one?
#   TUPLE=4,4,2 LS="48 64" bash tools/pmc_compare.sh        (separate --pmc passes per counter group; output: gpurun_out/pmc_compare.txt)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_compare; rm -rf $OUT; mkdir -p $OUT
TUPLE=${TUPLE:-4,4,2}; LS=${LS:-"48 64"}; LOG2N=${LOG2N:-28}
i=0
# (derived TCC_EA_*_sum counters are not available on gfx950 under rocprofv3 here: a pass that names them never returns — only these)
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU"; do
  i=$((i+1))
  for L in $LS; do
    timeout 120 rocprofv3 --pmc $P --kernel-include-regex "delta" -d $OUT/L$L/p$i -o pmc --output-format csv -- python3 tools/staircase.py $TUPLE $L $LOG2N 1 > $OUT/L$L.p$i.log 2>&1
  done
done
python3 - "$OUT" $LS <<'PY' | tee gpurun_out/pmc_compare.txt
import csv, glob, os, sys
out, Ls = sys.argv[1], sys.argv[2:]
table = {}
for L in Ls:
    for path in glob.glob(os.path.join(out, "L" + L, "p*", "**", "*counter_collection.csv"), recursive=True):
        sums, counts = {}, {}
        for r in csv.DictReader(open(path)):
            if "decompress" in r["Kernel_Name"]:
                c = r["Counter_Name"]; sums[c] = sums.get(c, 0.0) + float(r["Counter_Value"]); counts[c] = counts.get(c, 0) + 1
        for c in sums: table.setdefault(c, {})[L] = sums[c] / counts[c]
print("%-32s" % "counter (mean per launch)" + "".join("%18s" % ("L=" + L) for L in Ls))
for c in sorted(table):
    print("%-32s" % c + "".join("%18s" % ("-" if table[c].get(L) is None else "%.0f" % table[c][L]) for L in Ls))
for L in Ls:
    for line in open(os.path.join(out, "L%s.p1.log" % L)):
        if line.startswith("delta<"): print(line.strip())
PY
