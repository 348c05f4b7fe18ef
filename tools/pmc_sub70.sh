#!/bin/bash
# Runs ON THE GPU BOX: SQ counters per wavefront of the instantiations that sit below 70 % of peak in the sweep, beside a sibling that does not
# (profiles/r0N_pmc_below_70.txt; the cases: tools/pmc_sub70_cases.txt).  Three --pmc passes each (8 SQ slots per pass).
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/pmc_sub70; rm -rf $OUT; mkdir -p $OUT
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL"
run() {   # name, key pattern, kernel substring, log2n
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    rocprofv3 --pmc $P -d $OUT/$1/p$i -o pmc --output-format csv -- python3 tools/sweep_one.py "$2" $4 > $OUT/$1.p$i.log 2>&1
  done
  grep kernel_t $OUT/$1.p1.log | cut -c1-150
}
CASES=${CASES:-tools/pmc_sub70_cases.txt}
grep -v '^#' $CASES | while IFS='|' read name key kernel size; do run "$name" "$key" "$kernel" "$size"; done
python3 tools/pmc_sub70.py $OUT $CASES | tee gpurun_out/pmc_below_70.txt
