#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of one workload's decoder for several library variants (development aid).
#   tools/pmc_compare.sh WORKLOAD LOG2N KERNEL_SUBSTRING variant [variant ...]     ("default" = the in-tree library)
# Three --pmc passes per variant (8 SQ slots each), summarised per wavefront by tools/pmc_compare.py.
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
W=$1; N=$2; K=$3; shift 3
OUT=gpurun_out/pmc_compare/$W; rm -rf $OUT; mkdir -p $OUT
P1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES"
P2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_BUSY_CYCLES SQ_INSTS_BRANCH SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD"
for v in "$@"; do
  if [ "$v" = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    VERIFY=0 rocprofv3 --pmc $P -d $OUT/$v/p$i -o pmc --output-format csv -- python3 tools/bench_one.py $W $N 3 > $OUT/$v.p$i.log 2>&1
  done
  VERIFY=0 python3 tools/bench_one.py $W $N 10 | tail -1
done
python3 tools/pmc_compare.py $OUT "$K" "$@" | tee $OUT/summary.txt
