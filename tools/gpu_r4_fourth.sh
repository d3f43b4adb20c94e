#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_full_neighbourhood.py tests/test_gpu_validate.py tests/test_gpu_large.py -m gpu -x -q > gpurun_out/r4_prepush_tests.txt 2>&1
tail -5 gpurun_out/r4_prepush_tests.txt
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --regional --reps 3 base prepush=0 use_filters=1 prepush=1,max_sweeps=4 prepush=1,rounds_per_relabel=4 > gpurun_out/r4_prepush_ab512.jsonl 2>&1
cut -c1-420 gpurun_out/r4_prepush_ab512.jsonl
timeout 300 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu > gpurun_out/r4_bench_config3_a.json 2>gpurun_out/r4_bench_config3_a.err
cut -c1-1500 gpurun_out/r4_bench_config3_a.json; tail -3 gpurun_out/r4_bench_config3_a.err
