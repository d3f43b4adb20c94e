#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the 26-neighbourhood solve WITHOUT a regional term (k26_discharge, the workgroup form), a pass each
ROOT=${GRAFT_REPO_ROOT:-$PWD}; cd $ROOT; mkdir -p gpurun_out
OUT=$ROOT/gpurun_out/prof26n; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- python $ROOT/tools/gpu_ab.py --n 512 --conn 26 --reps 1 base > $OUT/fetch.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- python $ROOT/tools/gpu_ab.py --n 512 --conn 26 --reps 1 base > $OUT/write.log 2>&1
cd $ROOT
F=$(find $OUT/fetch -name "*.db" | head -1); W=$(find $OUT/write -name "*.db" | head -1)
[ -n "$F" ] && python tools/rocpd_summary.py pmc $F > gpurun_out/r4_26conn_noreg_fetch.csv
[ -n "$W" ] && python tools/rocpd_summary.py pmc $W > gpurun_out/r4_26conn_noreg_write.csv
rm -rf $OUT
head -6 gpurun_out/r4_26conn_noreg_fetch.csv; head -6 gpurun_out/r4_26conn_noreg_write.csv
