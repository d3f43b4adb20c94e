#!/bin/bash
# round-2 evidence: rocprofv3 passes of bench.py (headline) and of config 3, workloads, bench lines
ROOT=$GRAFT_REPO_ROOT; cd $ROOT; mkdir -p gpurun_out
bash tools/profile_round.sh r2
python bench.py 2>/dev/null | tail -1 > gpurun_out/r2_bench_n1.json
python bench.py --config 3 --no-cpu 2>/dev/null | tail -1 > gpurun_out/r2_bench_config3.json
python bench.py --config 2 --no-cpu 2>/dev/null | tail -1 > gpurun_out/r2_bench_config2.json
timeout 300 python tools/gpu_workloads.py 512 > gpurun_out/r2_workloads.jsonl 2>&1
# 26-neighbourhood: incremental relabel on / off with and without the regional term, and where k26_discharge's cycles go
timeout 300 python tools/gpu_26_incremental.py 512 sphere 2>&1 | grep incremental > gpurun_out/r2_workloads26.jsonl
timeout 200 python tools/gpu_sections26.py 512 1 2>&1 | grep -v Warn | tail -7 > gpurun_out/r2_discharge26_sections.txt
OUT=$ROOT/gpurun_out/prof26; rm -rf $OUT; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $ROOT/bench.py --config 3 --no-cpu --steps 2 > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- python $ROOT/bench.py --config 3 --no-cpu --steps 1 --warmup 0 > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- python $ROOT/bench.py --config 3 --no-cpu --steps 1 --warmup 0 > $OUT/write.log 2>&1
cd $ROOT
for k in trace fetch write; do D=$(find $OUT/$k -name "*.db" | head -1); [ -n "$D" ] && python tools/rocpd_summary.py $([ $k = trace ] && echo stats || echo pmc) $D > gpurun_out/r2_config3_$k.csv; rm -rf $OUT/$k; done
ls -la gpurun_out | head -40
