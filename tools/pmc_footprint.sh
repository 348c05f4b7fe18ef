#!/bin/bash
# Runs ON THE GPU BOX: memory-side counters (address translation, L2 <-> fabric latency and stalls) of one workload's decoder at several
# column sizes — does a decoder slow down with the FOOTPRINT (translation misses, longer fabric latency) or with something of its own?
#   tools/pmc_footprint.sh WORKLOAD KERNEL_SUBSTRING "28 31" [variant]
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
W=$1; K=$2; SIZES=$3; V=${4:-default}
if [ "$V" = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$V.so; fi
OUT=gpurun_out/pmc_footprint/$W; rm -rf $OUT; mkdir -p $OUT
P1="TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_TRANSLATION_MISS GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY"
P2="TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_EA0_WRREQ TCC_EA0_WRREQ_LEVEL"
P3="TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_WRREQ_STALL TCC_TAG_STALL"
P4="TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_WRITE_REQ TCP_PENDING_STALL_CYCLES"
P5="SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCP_UTCL1_STALL_INFLIGHT_MAX TCP_UTCL1_STALL_MULTI_MISS"
for s in $SIZES; do
  i=0
  for P in "$P1" "$P2" "$P3" "$P4" "$P5"; do
    i=$((i+1))
    VERIFY=0 rocprofv3 --pmc $P -d $OUT/$s/p$i -o pmc --output-format csv -- python3 tools/bench_one.py $W $s 3 > $OUT/$s.p$i.log 2>&1
  done
  VERIFY=0 python3 tools/bench_one.py $W $s 10 | tail -1 | cut -c1-150
done
python3 tools/pmc_footprint.py $OUT "$K" $SIZES | tee $OUT/summary.txt
