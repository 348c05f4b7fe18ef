cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=gpurun_out/r6d; rm -rf $OUT; mkdir -p $OUT
i=0
for P in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  for v in before_pipelined default; do
    if [ $v = default ]; then unset GIDDY_LIBRARY_PATH; else export GIDDY_LIBRARY_PATH=$PWD/variants/$v.so; fi
    timeout 300 rocprofv3 --pmc $P --kernel-include-regex "dictionary" -d $OUT/$v/p$i -o pmc --output-format csv -- python3 tools/sweep_one.py 'decompression::dictionary::kernel_t<4u, 8u, 2u>' 28 > $OUT/$v.p$i.log 2>&1
  done
done
unset GIDDY_LIBRARY_PATH
python3 tools/pmc_compare.py $OUT "dictionary::decompress" before_pipelined default | tee $OUT/compare.txt
