#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 600 python tools/gpu_wave_ab.py 512 ties 1:0:0 1:0:16 1:0:32 1:0:64 1:24:32 2>&1 | tee gpurun_out/ab_ties.jsonl
timeout 300 python tools/gpu_wave_ab.py 512 hard 1:0:0 1:0:12 1:0:16 1:16:12 2>&1 | tee gpurun_out/ab_hard.jsonl
