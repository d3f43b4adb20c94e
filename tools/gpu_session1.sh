#!/bin/bash
# round 3, GPU session 1: correctness of the folded small kernels + distance-transform relabel, their effect, kernel trace
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/s1_pytest.txt
timeout 300 python bench.py --no-cpu > gpurun_out/s1_bench.json 2> gpurun_out/s1_bench.err
timeout 600 python tools/gpu_ab.py --n 512 --tag s1 base first_relabel_dt=0 wave_kernels=8 > gpurun_out/s1_ab.jsonl 2>&1
timeout 300 python tools/gpu_ab.py --n 512 --wl hard --tag s1 base first_relabel_dt=0 >> gpurun_out/s1_ab.jsonl 2>&1
timeout 300 python tools/gpu_ab.py --n 256 --tag s1 base first_relabel_dt=0 >> gpurun_out/s1_ab.jsonl 2>&1
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/s1_trace -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $ROOT/gpurun_out/s1_trace.log 2>&1
cd $ROOT
T=$(find gpurun_out/s1_trace -name "*.db" | head -1)
[ -n "$T" ] && python tools/rocpd_summary.py stats $T > gpurun_out/s1_kernel_stats.csv
rm -rf gpurun_out/s1_trace
tail -3 gpurun_out/s1_pytest.txt; cat gpurun_out/s1_ab.jsonl | cut -c1-330
