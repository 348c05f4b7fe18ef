mkdir -p gpurun_out/r6h
timeout 1500 python -m pytest tests -m gpu -x -q -k "unaligned or delta or patched or sharding" > gpurun_out/r6h/gputests.log 2>&1; echo tests rc=$?; tail -5 gpurun_out/r6h/gputests.log
python tools/unaligned_bases.py 27 2>&1 | grep "delta" > gpurun_out/r6h/unaligned_delta.txt; cat gpurun_out/r6h/unaligned_delta.txt
python tools/fused_regret.py 28 --only "delta<4,2,1>|dictionary<4,8,2> 4096" --densities 0.1,0.2,0.5 2>&1 | tail -8
