#!/bin/bash
# round 3, GPU session 10: one list region (default) confirmed; config 3 kernel trace (regression hunt vs profiles/r2_config3_kernel_stats.csv)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/s10_ab.jsonl; : > $O
timeout 400 python tools/gpu_ab.py --n 512 --tag tree base >> $O 2>&1
timeout 300 python bench.py --no-cpu > gpurun_out/s10_bench.json 2> gpurun_out/s10_bench.err
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/s10_trace -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --config 3 > $ROOT/gpurun_out/s10_trace.log 2>&1
cd $ROOT
T=$(find gpurun_out/s10_trace -name "*.db" | head -1)
[ -n "$T" ] && python tools/rocpd_summary.py stats $T > gpurun_out/s10_config3_kernel_stats.csv
rm -rf gpurun_out/s10_trace
cut -c1-300 $O; head -16 gpurun_out/s10_config3_kernel_stats.csv | cut -c1-120; cut -c1-300 gpurun_out/s10_bench.json
