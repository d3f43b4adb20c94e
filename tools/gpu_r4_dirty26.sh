#!/bin/bash
# Finer DIRTY criterion of the 26-neighbourhood discharges: parity + what the incremental relabels cost now
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_full_neighbourhood.py -m gpu -x -q > gpurun_out/r4_dirty26_parity.txt 2>&1; tail -3 gpurun_out/r4_dirty26_parity.txt
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --reps 2 base trace=1 incremental_relabel=0 > gpurun_out/r4_dirty26.jsonl 2> gpurun_out/r4_dirty26_trace.txt
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --regional --reps 3 base trace=1 >> gpurun_out/r4_dirty26.jsonl 2> gpurun_out/r4_dirty26r_trace.txt
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --wl hard --reps 2 base >> gpurun_out/r4_dirty26.jsonl 2>/dev/null
cut -c1-420 gpurun_out/r4_dirty26.jsonl
grep "\[mgc\]" gpurun_out/r4_dirty26_trace.txt | head -40
grep "\[mgc\]" gpurun_out/r4_dirty26r_trace.txt | head -12
