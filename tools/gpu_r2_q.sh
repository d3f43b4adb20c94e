#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED|^ERROR" | tee gpurun_out/pytest_q.txt
timeout 600 python tools/gpu_r2_o.py 512 1024 2>&1 | grep "slab" | cut -c1-300 | tee gpurun_out/r2_slab_overhead_one_gpu.jsonl
