#!/bin/bash
# round-3 evidence on ONE MI355X (run through gpurun from the repo root): the whole GPU test tier (incl. the > 2^31-voxel volume),
# rocprofv3 passes of bench.py (headline) and of config 3, bench lines, workloads, section clocks, slab overhead.
ROOT=${GRAFT_REPO_ROOT:-$PWD}; cd $ROOT; mkdir -p gpurun_out; rm -f gpurun_out/parity_relaxations.jsonl
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > gpurun_out/r3_pytest.txt
bash tools/profile_round.sh r3
# every dispatch of one step in order (gaps between launches, the passes of a relabel one by one)
OUT=$ROOT/gpurun_out/prof_r3; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/tl -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu > $OUT/tl.log 2>&1
cd $ROOT; D=$(find $OUT/tl -name "*.db" | head -1); [ -n "$D" ] && python tools/rocpd_summary.py timeline $D > gpurun_out/r3_timeline.csv; rm -rf $OUT/tl
# roctx ranges of the solve (marker trace in a run of its own: no counters)
cd /tmp; MEDPY_HIP_ROCTX=1 timeout 300 rocprofv3 --marker-trace --kernel-trace --stats -d $OUT/mk -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu > $OUT/mk.log 2>&1
cd $ROOT; D=$(find $OUT/mk -name "*.db" | head -1)
[ -n "$D" ] && python tools/rocpd_summary.py ranges $D > gpurun_out/r3_roctx_ranges.csv
rm -rf $OUT/mk
python bench.py 2>gpurun_out/r3_bench.err | tail -1 > gpurun_out/r3_bench_n1.json
python bench.py --config 3 --no-cpu 2>>gpurun_out/r3_bench.err | tail -1 > gpurun_out/r3_bench_config3.json
python bench.py --config 2 --no-cpu 2>>gpurun_out/r3_bench.err | tail -1 > gpurun_out/r3_bench_config2.json
timeout 300 python tools/gpu_workloads.py 512 > gpurun_out/r3_workloads.jsonl 2>&1
MEDPY_HIP_LIB=$ROOT/build/lib_prof.so timeout 200 python tools/gpu_sections.py 512 2>&1 | grep -v Warn | tail -14 > gpurun_out/r3_discharge_sections.txt
timeout 300 python tools/gpu_slab_overhead.py 512 1024 > gpurun_out/r3_slab_overhead_one_gpu.jsonl 2>&1
# config 3 (26-neighbourhood + regional term): kernel trace
OUT=$ROOT/gpurun_out/prof26; rm -rf $OUT; mkdir -p $OUT; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $ROOT/bench.py --config 3 --no-cpu --steps 2 > $OUT/trace.log 2>&1
cd $ROOT; D=$(find $OUT/trace -name "*.db" | head -1); [ -n "$D" ] && python tools/rocpd_summary.py stats $D > gpurun_out/r3_config3_trace.csv; rm -rf $OUT
cat gpurun_out/r3_pytest.txt | tail -2; cut -c1-400 gpurun_out/r3_bench_n1.json; cat gpurun_out/r3_workloads.jsonl | cut -c1-200; head -5 gpurun_out/r3_roctx_ranges.csv
