#!/bin/bash
# Runs ON THE GPU BOX: SQ counters per wavefront of the instantiations that sit below 70 % of peak in the sweep, beside a sibling that does not
# (profiles/r02_pmc_below_70.txt).  Three --pmc passes each (8 SQ slots per pass).
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
run rle_8_1_8_8 'decompression::run_length_encoding::kernel_t<8u, 1u, 8u, 8u>' 'run_length_encoding::decompress<8u, 1u, 8u, 8u>' 30
run rle_4_1_1_4 'decompression::run_length_encoding::kernel_t<4u, 1u, 1u, 4u>' 'run_length_encoding::decompress<4u, 1u, 1u, 4u>' 30
run ib_4_1_2 'decompression::incidence_bitmaps::kernel_t<4u, 1u, 2u, true>' 'incidence_bitmaps::decompress<4u, 1u, 2u, true>' 28
run ib_4_4_2 'decompression::incidence_bitmaps::kernel_t<4u, 4u, 2u, true>' 'incidence_bitmaps::decompress<4u, 4u, 2u, true>' 28
run dzbv_4_2_8 'decompression::discard_zero_bytes::variable_width::kernel_t<4u, 2u, 8u>' 'variable_width::decompress<4u, 2u, 8u>' 28
run dzbv_4_4_8 'decompression::discard_zero_bytes::variable_width::kernel_t<4u, 4u, 8u>' 'variable_width::decompress<4u, 4u, 8u>' 28
run model_patched_short 'decompression::model::patched::kernel_t<4u, short, cuda::functors::unary::parametric_model::linear' 'model::patched::decompress<4u, short' 30
run model_patched_int 'decompression::model::patched::kernel_t<4u, int, cuda::functors::unary::parametric_model::linear' 'model::patched::decompress<4u, int' 30
python3 tools/pmc_sub70.py $OUT | tee gpurun_out/r02_pmc_below_70.txt
