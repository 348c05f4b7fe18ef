cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for M in 32 4; do
  export MEAN_RUN=$M
  mkdir -p variants; ln -sf ../libgiddy_amd/libgiddy_amd.so variants/m$M.so
done
MEAN_RUN=32 bash tools/pmc_compare.sh run_length_encoding 28 run_length_encoding::decompress m32 > /dev/null 2>&1; cp gpurun_out/pmc_compare/run_length_encoding/summary.txt gpurun_out/pmc_rle_m32.txt
MEAN_RUN=4 bash tools/pmc_compare.sh run_length_encoding 28 run_length_encoding::decompress m4 > /dev/null 2>&1; cp gpurun_out/pmc_compare/run_length_encoding/summary.txt gpurun_out/pmc_rle_m4.txt
paste gpurun_out/pmc_rle_m32.txt gpurun_out/pmc_rle_m4.txt | cut -c1-50,85-100
