#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert" | head -20 | tee gpurun_out/pytest_e.txt
timeout 300 python tools/gpu_wave_ab.py 512 sphere 1:0:0 1:0:0 2>&1 | tee gpurun_out/ab512.jsonl
timeout 300 python tools/gpu_wave_ab.py 256 sphere 1:0:0 0:0:0 2>&1 | tee gpurun_out/ab256.jsonl
timeout 300 python bench.py --no-cpu 2>&1 | tail -1 | tee gpurun_out/bench_e.json
