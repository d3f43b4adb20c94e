#!/bin/bash
# Finer DIRTY criterion, 6-neighbourhood kernels: parity + headline and workloads A/B
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_validate.py tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/r4_dirty6_parity.txt 2>&1; tail -3 gpurun_out/r4_dirty6_parity.txt
timeout 600 python tools/gpu_ab.py --n 512 --reps 5 base trace=1 > gpurun_out/r4_dirty6.jsonl 2> gpurun_out/r4_dirty6_trace.txt
timeout 600 python tools/gpu_ab.py --n 512 --regional --reps 3 base >> gpurun_out/r4_dirty6.jsonl 2>/dev/null
timeout 600 python tools/gpu_ab.py --n 512 --wl hard --reps 3 base >> gpurun_out/r4_dirty6.jsonl 2>/dev/null
timeout 600 python tools/gpu_ab.py --n 512 --wl ties --reps 2 base >> gpurun_out/r4_dirty6.jsonl 2>/dev/null
timeout 600 python tools/gpu_ab.py --n 256 --reps 5 base >> gpurun_out/r4_dirty6.jsonl 2>/dev/null
cut -c1-420 gpurun_out/r4_dirty6.jsonl
grep "\[mgc\]" gpurun_out/r4_dirty6_trace.txt | head -24
python bench.py > gpurun_out/r4_dirty6_bench.json 2>/dev/null; cut -c1-600 gpurun_out/r4_dirty6_bench.json
