#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert|^E " | head -30 | tee gpurun_out/pytest_f.txt
timeout 300 python tools/gpu_wave_ab.py 512 sphere 1:0:0 2>&1 | tee gpurun_out/ab512.jsonl
