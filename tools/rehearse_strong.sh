#!/bin/bash
# Runs ON THE GPU BOX: rehearses bench.py's N > 1 flow on ONE GPU (every rank on device 0, gloo instead of RCCL):
# the default weak line with its `sharded` strong-scaling entries, then --scaling strong as the headline.
cd "$GRAFT_REPO_ROOT"
export GIDDY_BENCH_SHARE_DEVICE=1 GIDDY_BENCH_BACKEND=gloo
W=${1:-2}
python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus $W --steps 5 --warmup 1 --log2n 28 --no-cpu-baseline --schemes none
python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus $W --steps 5 --warmup 1 --log2n 28 --scaling strong --workload run_position_encoding_absolute --gather
