mkdir -p gpurun_out/r6c
python tools/fused_regret.py 28 > gpurun_out/r6c/fused_regret.txt 2>&1
python tools/unaligned_bases.py 27 > gpurun_out/r6c/unaligned_bases.txt 2>&1
bash tools/sweep_variants.sh 'dictionary::kernel_t<.u, 8u, 2u>|dictionary::kernel_t<4u, 4u, 1u>' 28 before_pipelined default before_pipelined default > gpurun_out/r6c/sweep_dictionary.txt 2>&1
tail -3 gpurun_out/r6c/fused_regret.txt
