#!/bin/bash
# rocprofv3 passes whose summaries go to profiles/ (run on the GPU box through gpurun from the repo root):
#   kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own passes (counters never share a run with traces
#   other than --kernel-trace).  Usage: bash tools/profile_round.sh <tag>
set -u
TAG=${1:-r1}
ROOT=$PWD
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-extras > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-extras > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-extras > $OUT/write.log 2>&1
cd $ROOT
T=$(find $OUT/trace -name "*.db" | head -1); F=$(find $OUT/fetch -name "*.db" | head -1); W=$(find $OUT/write -name "*.db" | head -1)
python tools/rocpd_summary.py stats $T > gpurun_out/${TAG}_kernel_stats.csv
python tools/rocpd_summary.py pmc $F > gpurun_out/${TAG}_pmc_fetch_size.csv
python tools/rocpd_summary.py pmc $W > gpurun_out/${TAG}_pmc_write_size.csv
python tools/rocpd_summary.py json $F $W > gpurun_out/pmc_discharge.json
rm -rf $OUT/trace $OUT/fetch $OUT/write
# instruction mix of the solver kernels (one pass per counter group; counters never share a run with traces other than --kernel-trace)
cd /tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  N=$(echo $C | tr ' ' '_')
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d $OUT/$N -- python $ROOT/bench.py --steps 1 --warmup 0 --no-cpu --no-extras > $OUT/$N.log 2>&1
  D=$(find $OUT/$N -name "*.db" | head -1)
  [ -n "$D" ] && python $ROOT/tools/rocpd_summary.py pmc $D > $ROOT/gpurun_out/${TAG}_pmc_$N.csv
  rm -rf $OUT/$N
done
cd $ROOT
