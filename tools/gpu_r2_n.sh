#!/bin/bash
# the multi-GPU bench code path at its real per-rank size, two ranks sharing the one GPU of this box (development transport)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export MEDPY_DIST_BACKEND=gloo MEDPY_BENCH_ANY_WORLD=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29721 bench.py --gpus 2 --steps 1 --warmup 1 2>&1 | grep "^{" | tee gpurun_out/r2_bench_2ranks_shared_gpu.json | cut -c1-1500
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29722 bench.py --gpus 2 --config 5 --steps 1 --warmup 0 2>&1 | grep "^{" | tee gpurun_out/r2_bench_config5_2ranks_shared_gpu.json | cut -c1-1500
