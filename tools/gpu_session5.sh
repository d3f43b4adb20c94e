#!/bin/bash
# round 3, GPU session 5: sharded list counters / tickets, bulk activation; tail order and prefetch A/B on top
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/s5_ab.jsonl; : > $O; rm -f gpurun_out/parity_relaxations.jsonl
( MEDPY_SKIP_BIG_IDS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_slabs.py tests/test_gpu_large.py tests/test_gpu_validate.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/s5_pytest.txt
timeout 400 python tools/gpu_ab.py --n 512 --tag tree base wave_stagger=3 first_relabel_dt=0 >> $O 2>&1
for V in pf2; do
  [ -f build/lib_$V.so ] || continue
  timeout 200 python tools/gpu_ab.py --n 512 --tag $V --lib $PWD/build/lib_$V.so base >> $O 2>&1
done
timeout 200 python tools/gpu_ab.py --n 512 --wl hard --tag tree base >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 256 --tag tree base >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 128 --tag tree base >> $O 2>&1
timeout 300 python bench.py --no-cpu > gpurun_out/s5_bench.json 2> gpurun_out/s5_bench.err
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/s5_trace -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $ROOT/gpurun_out/s5_trace.log 2>&1
cd $ROOT
T=$(find gpurun_out/s5_trace -name "*.db" | head -1)
[ -n "$T" ] && python tools/rocpd_summary.py stats $T > gpurun_out/s5_kernel_stats.csv
rm -rf gpurun_out/s5_trace
tail -3 gpurun_out/s5_pytest.txt; cut -c1-300 $O
