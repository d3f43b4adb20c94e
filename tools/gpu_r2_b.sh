#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== ab 128" ; timeout 300 python tools/gpu_wave_ab.py 128 sphere 0:0:0 1:0:0 2:0:0 3:0:0 7:0:0 2>&1 | tee gpurun_out/ab128.jsonl
echo "== ab 512" ; timeout 600 python tools/gpu_wave_ab.py 512 sphere 0:0:0 1:0:0 2:0:0 3:0:0 7:0:0 5:0:0 3:8:0 3:16:0 3:0:6 3:0:10 1:0:0:1024 1:0:0:4096 2>&1 | tee gpurun_out/ab512.jsonl
echo "== pytest" ; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/pytest_b.txt
