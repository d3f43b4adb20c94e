#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert|^E " | head -30 | tee gpurun_out/pytest_k.txt
timeout 600 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_k_n1.json
timeout 300 python bench.py --config 3 --no-cpu 2>&1 | tail -1 | tee gpurun_out/bench_k_config3.json
timeout 300 python bench.py --config 2 --no-cpu 2>&1 | tail -1 | tee gpurun_out/bench_k_config2.json
