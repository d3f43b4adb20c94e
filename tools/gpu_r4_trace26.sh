#!/bin/bash
# Per-relabel tile counts of the 26-neighbourhood solve without a regional term (where do the relabel visits go?)
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --reps 1 trace=1 > gpurun_out/r4_trace26.jsonl 2> gpurun_out/r4_trace26.txt
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --regional --reps 1 trace=1 > gpurun_out/r4_trace26r.jsonl 2> gpurun_out/r4_trace26r.txt
grep "\[mgc\]" gpurun_out/r4_trace26.txt | head -60
grep "\[mgc\]" gpurun_out/r4_trace26r.txt | head -30
