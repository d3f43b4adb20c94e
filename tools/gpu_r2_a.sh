#!/bin/bash
# round 2, first GPU pass: wave kernels -- correctness A/B, full GPU test-suite, 512^3 A/B, bench, kernel trace
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== ab 128" ; timeout 300 python tools/gpu_wave_ab.py 128 sphere 0:0:0 1:0:0 2:0:0 3:0:0 7:0:0 2>&1 | tee gpurun_out/ab128.jsonl
echo "== ab 256" ; timeout 300 python tools/gpu_wave_ab.py 256 sphere 0:0:0 3:0:0 7:0:0 2>&1 | tee gpurun_out/ab256.jsonl
echo "== pytest" ; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/pytest_a.txt
echo "== ab 512" ; timeout 600 python tools/gpu_wave_ab.py 512 sphere 0:0:0 1:0:0 2:0:0 3:0:0 7:0:0 3:8:0 3:16:0 7:8:0 3:0:6 3:0:10 7:0:10 2>&1 | tee gpurun_out/ab512.jsonl
echo "== bench" ; timeout 600 python bench.py 2>&1 | tail -3 | tee gpurun_out/bench_a.json
