#!/bin/bash
# round 3, GPU session 12: incremental relabels over bricks of 2 x 2 x 2 tiles vs over tiles
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/s12_ab.jsonl; : > $O; rm -f gpurun_out/parity_relaxations.jsonl
( MEDPY_SKIP_BIG_IDS=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/s12_pytest.txt
timeout 400 python tools/gpu_ab.py --n 512 --tag tree base relabel_bricks=0 base relabel_bricks=0 >> $O 2>&1
timeout 300 python tools/gpu_ab.py --n 512 --wl hard --tag tree base relabel_bricks=0 >> $O 2>&1
timeout 400 python tools/gpu_ab.py --n 512 --wl ties --reps 2 --tag tree base relabel_bricks=0 >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 256 --tag tree base relabel_bricks=0 >> $O 2>&1
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/s12_trace -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $ROOT/gpurun_out/s12_trace.log 2>&1
cd $ROOT
T=$(find gpurun_out/s12_trace -name "*.db" | head -1)
[ -n "$T" ] && python tools/rocpd_summary.py stats $T > gpurun_out/s12_kernel_stats.csv
rm -rf gpurun_out/s12_trace
tail -3 gpurun_out/s12_pytest.txt; python - <<'P'
import json
for l in open('gpurun_out/s12_ab.jsonl'):
    if not l.startswith('{'): print(l.strip()[:200]); continue
    d=json.loads(l); print(d['wl'],d['n'],d['variant'],d['ms'],'dis',d['discharge_ms'],'rel',d['relabel_ms'],'rb',d['readbacks'],'rl',d['rel_launches'],'rt',d['rel_tiles'],'same',d['same_labels'])
P
head -8 gpurun_out/s12_kernel_stats.csv | cut -c1-120
