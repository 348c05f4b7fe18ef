mkdir -p gpurun_out/r6j
timeout 1500 python -m pytest tests -m gpu -x -q -k "patched or boundary" > gpurun_out/r6j/gputests.log 2>&1; echo tests rc=$?; tail -2 gpurun_out/r6j/gputests.log
python tools/fused_regret.py 28 --parts --only "delta<4,2,1>|dictionary<4,8,2> 4096|model<4,2" --densities 0.001,0.01,0.2,0.5 2>&1 | tail -14 | cut -c1-260
bash tools/sweep_variants.sh 'decompression::dictionary::kernel_t' 28 steps8 default steps8 default > gpurun_out/r6j/sweep.txt 2>&1
python tools/compare_sweeps.py gpurun_out/r6j/sweep.txt default steps8 | cut -c1-200
