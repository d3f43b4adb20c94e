#!/bin/bash
# workloads + config 3 (26-conn + regional) with rocprofv3 kernel stats and HBM counters
mkdir -p gpurun_out; ROOT=$GRAFT_REPO_ROOT; cd $ROOT
export TMPDIR=/tmp
timeout 300 python tools/gpu_workloads.py 512 2>&1 | tee gpurun_out/r2_workloads.jsonl
timeout 300 python tools/gpu_probe26.py 512 2>&1 | tee gpurun_out/r2_config3_512.jsonl
OUT=$ROOT/gpurun_out/prof26; mkdir -p $OUT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $ROOT/tools/gpu_probe26.py 512 > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- python $ROOT/tools/gpu_probe26.py 512 > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- python $ROOT/tools/gpu_probe26.py 512 > $OUT/write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU -d $OUT/inst -- python $ROOT/tools/gpu_probe26.py 512 > $OUT/inst.log 2>&1
cd $ROOT
for k in trace fetch write inst; do D=$(find $OUT/$k -name "*.db" | head -1); [ -n "$D" ] && python tools/rocpd_summary.py $([ $k = trace ] && echo stats || echo pmc) $D > gpurun_out/r2_config3_$k.csv; rm -rf $OUT/$k; done
