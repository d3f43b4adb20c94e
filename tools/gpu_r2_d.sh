#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/gpu_wave_ab.py 512 sphere 1:0:0 5:0:0 1:8:0 1:16:0 1:0:10 1:0:6 2>&1 | tee gpurun_out/ab512.jsonl
timeout 300 python tools/gpu_wave_ab.py 256 sphere 0:0:0 1:0:0 5:0:0 2>&1 | tee gpurun_out/ab256.jsonl
timeout 300 python tools/gpu_wave_ab.py 128 sphere 0:0:0 1:0:0 2>&1 | tee gpurun_out/ab128.jsonl
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tee gpurun_out/pytest_d.txt
