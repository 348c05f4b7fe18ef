mkdir -p gpurun_out/r6e
timeout 1200 python -m pytest tests -m gpu -x -q -k "dictionary or patched or fixed or frame or examples or headline or golden" > gpurun_out/r6e/gputests.log 2>&1; echo tests rc=$?; tail -2 gpurun_out/r6e/gputests.log
bash tools/sweep_variants.sh 'dictionary|fixed_width::patched|frame_of_reference::patched' 28 before_pipelined default before_pipelined default > gpurun_out/r6e/sweep.txt 2>&1
python tools/compare_sweeps.py gpurun_out/r6e/sweep.txt before_pipelined default > gpurun_out/r6e/compare.txt; cat gpurun_out/r6e/compare.txt | cut -c1-200
