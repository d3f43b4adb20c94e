#!/bin/bash
# round 3, GPU session 14: k_build with one vote barrier and t-link planes only where needed; relabel pass with a speculative first entry
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/s14_ab.jsonl; : > $O; rm -f gpurun_out/parity_relaxations.jsonl
( MEDPY_SKIP_BIG_IDS=1 timeout 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/s14_pytest.txt
timeout 300 python tools/gpu_ab.py --n 512 --tag s14 base max_sweeps=6 max_sweeps=12 base >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 512 --wl hard --tag s14 base >> $O 2>&1
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/s14_trace -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $ROOT/gpurun_out/s14_trace.log 2>&1
cd $ROOT
T=$(find gpurun_out/s14_trace -name "*.db" | head -1)
[ -n "$T" ] && python tools/rocpd_summary.py stats $T > gpurun_out/s14_kernel_stats.csv
rm -rf gpurun_out/s14_trace
timeout 300 python bench.py --no-cpu > gpurun_out/s14_bench.json 2> gpurun_out/s14_bench.err
timeout 300 python bench.py --no-cpu --config 3 > gpurun_out/s14_bench_config3.json 2>> gpurun_out/s14_bench.err
tail -3 gpurun_out/s14_pytest.txt; python - <<'P'
import json
for l in open('gpurun_out/s14_ab.jsonl'):
    if not l.startswith('{'): print(l.strip()[:200]); continue
    d=json.loads(l); print(d['wl'],d['n'],d['variant'],d['ms'],'build',d['build_ms'],'dis',d['discharge_ms'],'rel',d['relabel_ms'],'rb',d['readbacks'],'dl',d['dis_launches'],'rl',d['rel_launches'],'dt',d['dis_tiles'],'same',d['same_labels'])
P
head -8 gpurun_out/s14_kernel_stats.csv | cut -c1-120
cut -c1-200 gpurun_out/s14_bench.json; cut -c1-200 gpurun_out/s14_bench_config3.json
