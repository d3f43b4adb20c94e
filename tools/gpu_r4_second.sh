#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/r4_counters_avail.txt 2>&1
MEDPY_HIP_LIB=build/lib_prof26.so timeout 300 python tools/gpu_sections26w.py 512 regional wave_kernels=41 > gpurun_out/r4_w26_sections.txt 2>&1
MEDPY_HIP_LIB=build/lib_prof26.so timeout 300 python tools/gpu_sections26w.py 512 regional wave_kernels=41,wave_grid26=256 > gpurun_out/r4_w26_sections_g256.txt 2>&1
cat gpurun_out/r4_w26_sections.txt gpurun_out/r4_w26_sections_g256.txt
cd /tmp
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_IFETCH SQ_IFETCH_LEVEL" "SQC_ICACHE_REQ SQC_ICACHE_MISSES"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$N -- python $GRAFT_REPO_ROOT/tools/gpu_ab.py --n 256 --conn 26 --regional --reps 1 wave_kernels=41 > /tmp/pmc_$N.log 2>&1
  D=$(find /tmp/pmc_$N -name "*.db" | head -1)
  [ -n "$D" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py pmc $D > $GRAFT_REPO_ROOT/gpurun_out/r4_w26_pmc_$N.csv 2>&1
  tail -3 /tmp/pmc_$N.log
done
cd $GRAFT_REPO_ROOT
head -30 gpurun_out/r4_w26_pmc_*.csv | cut -c1-300
