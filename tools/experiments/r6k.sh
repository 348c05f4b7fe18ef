mkdir -p gpurun_out/r6k
timeout 1500 python -m pytest tests -m gpu -x -q -k "patched or dictionary or scatter or batch or sharding" > gpurun_out/r6k/gputests.log 2>&1; echo tests rc=$?; tail -2 gpurun_out/r6k/gputests.log
python tools/fused_regret.py 28 --parts --only "delta<4,2,1>|dictionary<4,8,2> 4096|model<4,2" --densities 0.2,0.5 2>&1 | tail -8 | cut -c1-260
bash tools/sweep_variants.sh 'decompression::dictionary::kernel_t' 28 before_pipelined default before_pipelined default > gpurun_out/r6k/sweep.txt 2>&1
python tools/compare_sweeps.py gpurun_out/r6k/sweep.txt before_pipelined default | cut -c1-200
