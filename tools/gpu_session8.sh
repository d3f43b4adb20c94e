#!/bin/bash
# round 3, GPU session 8: wave-form absorb, alternating filter slots, adaptive relabel batches; list shards 1 vs 16; resident waves
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/s8_ab.jsonl; : > $O; rm -f gpurun_out/parity_relaxations.jsonl
( MEDPY_SKIP_BIG_IDS=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/s8_pytest.txt
timeout 400 python tools/gpu_ab.py --n 512 --tag tree base list_shards=1 wave_grid_dis=1024 wave_grid_dis=1536 wave_grid_dis=4096 >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 512 --wl hard --tag tree base list_shards=1 >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 256 --tag tree base list_shards=1 >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 128 --tag tree base >> $O 2>&1
timeout 300 python bench.py --no-cpu > gpurun_out/s8_bench.json 2> gpurun_out/s8_bench.err
timeout 300 python bench.py --no-cpu --config 3 > gpurun_out/s8_bench_config3.json 2>> gpurun_out/s8_bench.err
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/s8_trace -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $ROOT/gpurun_out/s8_trace.log 2>&1
cd $ROOT
T=$(find gpurun_out/s8_trace -name "*.db" | head -1)
[ -n "$T" ] && python tools/rocpd_summary.py stats $T > gpurun_out/s8_kernel_stats.csv
rm -rf gpurun_out/s8_trace
tail -3 gpurun_out/s8_pytest.txt; cut -c1-300 $O; head -30 gpurun_out/s8_kernel_stats.csv | cut -c1-110; cut -c1-400 gpurun_out/s8_bench_config3.json
