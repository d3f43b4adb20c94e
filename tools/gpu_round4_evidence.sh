#!/bin/bash
# round-4 evidence on ONE MI355X (run through gpurun from the repo root): the whole GPU test tier, rocprofv3 passes of bench.py
# (headline) and of --config 3 (kernel trace + FETCH_SIZE / WRITE_SIZE passes for k26_discharge), bench lines, workloads.
ROOT=${GRAFT_REPO_ROOT:-$PWD}; cd $ROOT; mkdir -p gpurun_out; rm -f gpurun_out/parity_relaxations.jsonl
( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > gpurun_out/r4_pytest.txt
bash tools/profile_round.sh r4
OUT=$ROOT/gpurun_out/prof_r4; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/tl -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu > $OUT/tl.log 2>&1
cd $ROOT; D=$(find $OUT/tl -name "*.db" | head -1); [ -n "$D" ] && python tools/rocpd_summary.py timeline $D > gpurun_out/r4_timeline.csv; rm -rf $OUT/tl
# config 3 (26-neighbourhood + regional term): kernel trace, then FETCH_SIZE and WRITE_SIZE in passes of their own
OUT=$ROOT/gpurun_out/prof26; rm -rf $OUT; mkdir -p $OUT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $ROOT/bench.py --config 3 --no-cpu --steps 2 --warmup 1 > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- python $ROOT/bench.py --config 3 --no-cpu --steps 1 --warmup 0 > $OUT/fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- python $ROOT/bench.py --config 3 --no-cpu --steps 1 --warmup 0 > $OUT/write.log 2>&1
cd $ROOT
T=$(find $OUT/trace -name "*.db" | head -1); F=$(find $OUT/fetch -name "*.db" | head -1); W=$(find $OUT/write -name "*.db" | head -1)
[ -n "$T" ] && python tools/rocpd_summary.py stats $T > gpurun_out/r4_config3_trace.csv
[ -n "$F" ] && python tools/rocpd_summary.py pmc $F > gpurun_out/r4_config3_fetch.csv
[ -n "$W" ] && python tools/rocpd_summary.py pmc $W > gpurun_out/r4_config3_write.csv
[ -n "$F" ] && [ -n "$W" ] && python tools/rocpd_summary.py json $F $W k26_discharge_w > gpurun_out/pmc_discharge26.json
rm -rf $OUT
# the bench lines read the PMC summaries from profiles/: put the fresh ones there first (same sources: the hash inside says so)
cp gpurun_out/pmc_discharge.json gpurun_out/pmc_discharge26.json profiles/ 2>/dev/null
python bench.py 2>gpurun_out/r4_bench.err | tail -1 > gpurun_out/r4_bench_n1.json
python bench.py --config 3 --no-cpu 2>>gpurun_out/r4_bench.err | tail -1 > gpurun_out/r4_bench_config3.json
python bench.py --config 2 --no-cpu 2>>gpurun_out/r4_bench.err | tail -1 > gpurun_out/r4_bench_config2.json
timeout 300 python tools/gpu_workloads.py 512 > gpurun_out/r4_workloads.jsonl 2>&1
( timeout 300 python tools/gpu_ab.py --n 512 --conn 26 --reps 2 base; timeout 300 python tools/gpu_ab.py --n 512 --conn 26 --regional --reps 2 base prepush=0; timeout 300 python tools/gpu_ab.py --n 512 --conn 26 --wl hard --reps 2 base ) > gpurun_out/r4_workloads26.jsonl 2>&1
OUT=$ROOT/gpurun_out/prof26n; rm -rf $OUT; mkdir -p $OUT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $ROOT/tools/gpu_ab.py --n 512 --conn 26 --reps 1 base > $OUT/trace.log 2>&1
cd $ROOT; T=$(find $OUT/trace -name "*.db" | head -1); [ -n "$T" ] && python tools/rocpd_summary.py stats $T > gpurun_out/r4_26conn_noreg_trace.csv; rm -rf $OUT
( timeout 300 python tools/gpu_ab.py --n 512 --regional --reps 3 base prepush=0 ) > gpurun_out/r4_regional_6conn.jsonl 2>&1
( timeout 300 python tools/gpu_ab.py --n 512 --conn 26 --reps 1 trace=1 ) > /dev/null 2> gpurun_out/r4_relabel_trace26.txt
MEDPY_HIP_LIB=$ROOT/build/lib_prof.so timeout 200 python tools/gpu_sections26.py 512 0 2>&1 | grep -v Warn | tail -6 > gpurun_out/r4_discharge26_sections_noreg.txt
MEDPY_HIP_LIB=$ROOT/build/lib_prof.so timeout 200 python tools/gpu_sections.py 512 2>&1 | grep -v Warn | tail -14 > gpurun_out/r4_discharge_sections.txt
cat gpurun_out/r4_pytest.txt | tail -2; cut -c1-600 gpurun_out/r4_bench_n1.json; cut -c1-300 gpurun_out/r4_bench_config3.json; cut -c1-200 gpurun_out/r4_workloads.jsonl; cut -c1-300 gpurun_out/r4_workloads26.jsonl
