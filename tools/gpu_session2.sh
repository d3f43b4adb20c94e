#!/bin/bash
# round 3, GPU session 2: discharge tail reorder + run-ahead, DPP x-shifts, prefetch levels, status-word activation, label summaries
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/s2_ab.jsonl; : > $O
( MEDPY_SKIP_BIG_IDS=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/s2_pytest.txt
timeout 400 python tools/gpu_ab.py --n 512 --tag tree base wave_stagger=1 wave_stagger=3 activate_exact_max=1000000 activate_exact_max=0 max_sweeps=16 rounds_per_relabel=10 >> $O 2>&1
for V in pf1 pf2 nodpp; do
  [ -f build/lib_$V.so ] || continue
  timeout 200 python tools/gpu_ab.py --n 512 --tag $V --lib $PWD/build/lib_$V.so base >> $O 2>&1
done
timeout 200 python tools/gpu_ab.py --n 512 --wl hard --tag tree base >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 512 --wl hard --tag pf2 --lib $PWD/build/lib_pf2.so base >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 256 --tag tree base >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 128 --tag tree base >> $O 2>&1
timeout 300 python bench.py --no-cpu > gpurun_out/s2_bench.json 2> gpurun_out/s2_bench.err
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/s2_trace -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $ROOT/gpurun_out/s2_trace.log 2>&1
cd $ROOT
T=$(find gpurun_out/s2_trace -name "*.db" | head -1)
[ -n "$T" ] && python tools/rocpd_summary.py stats $T > gpurun_out/s2_kernel_stats.csv
rm -rf gpurun_out/s2_trace
tail -3 gpurun_out/s2_pytest.txt; cut -c1-300 $O
