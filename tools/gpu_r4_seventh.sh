#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4_pytest_a.txt 2>&1
tail -3 gpurun_out/r4_pytest_a.txt
timeout 300 python bench.py --config 3 --steps 5 --warmup 1 --no-cpu > gpurun_out/r4_bench_config3_b.json 2>gpurun_out/r4_bench_config3_b.err
cut -c1-300 gpurun_out/r4_bench_config3_b.json
timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/r4_bench_n1_a.json 2>/dev/null
cut -c1-300 gpurun_out/r4_bench_n1_a.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt3 -- python $GRAFT_REPO_ROOT/bench.py --config 3 --steps 3 --warmup 1 --no-cpu > /tmp/kt3.log 2>&1
D=$(find /tmp/kt3 -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $D > $GRAFT_REPO_ROOT/gpurun_out/r4_config3_trace_a.csv 2>&1
head -24 $GRAFT_REPO_ROOT/gpurun_out/r4_config3_trace_a.csv | cut -c1-200
timeout 300 python $GRAFT_REPO_ROOT/tools/gpu_ab.py --n 512 --conn 26 --reps 2 base > $GRAFT_REPO_ROOT/gpurun_out/r4_noreg26.jsonl 2>&1
cut -c1-400 $GRAFT_REPO_ROOT/gpurun_out/r4_noreg26.jsonl
