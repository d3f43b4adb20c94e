#!/bin/bash
# round 3, GPU session 23: dispatch timeline of one bench step (where the passes of a relabel spend their time, gaps between launches)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/s23_trace -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu > $ROOT/gpurun_out/s23_trace.log 2>&1
cd $ROOT
T=$(find gpurun_out/s23_trace -name "*.db" | head -1)
[ -n "$T" ] && python tools/rocpd_summary.py timeline $T > gpurun_out/s23_timeline.csv
rm -rf gpurun_out/s23_trace
wc -l gpurun_out/s23_timeline.csv; tail -3 gpurun_out/s23_trace.log | cut -c1-200
