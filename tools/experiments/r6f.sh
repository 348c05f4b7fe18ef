mkdir -p gpurun_out/r6f
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r6f/gputests.log 2>&1; echo tests rc=$?; tail -3 gpurun_out/r6f/gputests.log
python tools/fused_regret.py 28 > gpurun_out/r6f/fused_regret.txt 2>&1
grep -v "regret  0.0\|regret  [0-2]\.[0-9]" gpurun_out/r6f/fused_regret.txt | cut -c1-170
bash tools/sweep_variants.sh 'dictionary::kernel_t<.u, 8u|dictionary::patched::kernel_t<.u, 8u, 8u>|dictionary::patched::kernel_t<4u, 4u, 1u>' 28 before_pipelined default before_pipelined default > gpurun_out/r6f/sweep.txt 2>&1
python tools/compare_sweeps.py gpurun_out/r6f/sweep.txt before_pipelined default | cut -c1-200
