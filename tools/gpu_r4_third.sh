#!/bin/bash
set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
MEDPY_HIP_LIB=build/lib_prof26.so timeout 300 python tools/gpu_sections26w.py 512 regional wave_kernels=41 > gpurun_out/r4_w26_sections_b.txt 2>&1
cat gpurun_out/r4_w26_sections_b.txt
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --regional --reps 2 base wave_kernels=41 wave_kernels=41,w26_passes=1 > gpurun_out/r4_w26_ab512_b.jsonl 2>&1
cut -c1-420 gpurun_out/r4_w26_ab512_b.jsonl
cd /tmp; export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS"; do
  N=$(echo $C | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmcb_$N -- python $GRAFT_REPO_ROOT/tools/gpu_ab.py --n 256 --conn 26 --regional --reps 1 base > /tmp/pmcb_$N.log 2>&1
  D=$(find /tmp/pmcb_$N -name "*.db" | head -1)
  [ -n "$D" ] && python $GRAFT_REPO_ROOT/tools/rocpd_summary.py pmc $D | grep -i "k26_discharge\|^kernel" > $GRAFT_REPO_ROOT/gpurun_out/r4_k26base_pmc_$N.csv 2>&1
done
cd $GRAFT_REPO_ROOT; cat gpurun_out/r4_k26base_pmc_*.csv | cut -c1-200
