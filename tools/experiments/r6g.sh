mkdir -p gpurun_out/r6g
timeout 1500 python -m pytest tests -m gpu -x -q -k "patched or scatter or dictionary" > gpurun_out/r6g/gputests.log 2>&1; echo tests rc=$?; tail -2 gpurun_out/r6g/gputests.log
python tools/fused_regret.py 28 > gpurun_out/r6g/fused_regret.txt 2>&1
grep -v "regret  0.0\|regret  [0-2]\.[0-9]" gpurun_out/r6g/fused_regret.txt | cut -c1-170
bash tools/sweep_variants.sh 'fixed_width::patched|frame_of_reference::patched|dictionary::patched' 28 before_pipelined default before_pipelined default > gpurun_out/r6g/sweep.txt 2>&1
python tools/compare_sweeps.py gpurun_out/r6g/sweep.txt before_pipelined default | cut -c1-200
