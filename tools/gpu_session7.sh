#!/bin/bash
# round 3, GPU session 7: run-ahead off (default) vs on; eight-section profile of the wave discharge
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/s7_ab.jsonl; : > $O
( MEDPY_SKIP_BIG_IDS=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "synthetic or golden or config2" 2>&1 | tail -5 ) > gpurun_out/s7_pytest.txt
timeout 400 python tools/gpu_ab.py --n 512 --tag tree base wave_stagger=3 >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 512 --tag ra --lib $PWD/build/lib_ra.so base >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 512 --wl hard --tag tree base >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 256 --tag tree base >> $O 2>&1
MEDPY_HIP_LIB=$PWD/build/lib_prof.so timeout 300 python tools/gpu_sections.py 512 > gpurun_out/s7_sections.txt 2>&1
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/s7_trace -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $ROOT/gpurun_out/s7_trace.log 2>&1
cd $ROOT
T=$(find gpurun_out/s7_trace -name "*.db" | head -1)
[ -n "$T" ] && python tools/rocpd_summary.py stats $T > gpurun_out/s7_kernel_stats.csv
rm -rf gpurun_out/s7_trace
tail -3 gpurun_out/s7_pytest.txt; cut -c1-300 $O; cat gpurun_out/s7_sections.txt; head -8 gpurun_out/s7_kernel_stats.csv
