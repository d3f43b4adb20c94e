#!/bin/bash
# round 3, GPU session 15: the passes of a global relabel as one launch over a queue (k_relabel_q)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/s15_ab.jsonl; : > $O
timeout 300 python tools/gpu_ab.py --n 256 --tag s15 base relabel_queue=1 >> $O 2>&1
timeout 300 python tools/gpu_ab.py --n 512 --tag s15 base relabel_queue=1 relabel_queue=1,queue_grid=1024 relabel_queue=1,queue_grid=4096 base >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 512 --wl hard --tag s15 base relabel_queue=1 >> $O 2>&1
( MEDPY_HIP_PARAMS=relabel_queue=1 MEDPY_SKIP_BIG_IDS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py tests/test_gpu_edge_cases.py tests/test_gpu_validate.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/s15_pytest.txt
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
MEDPY_HIP_PARAMS=relabel_queue=1 timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/s15_trace -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $ROOT/gpurun_out/s15_trace.log 2>&1
cd $ROOT
T=$(find gpurun_out/s15_trace -name "*.db" | head -1)
[ -n "$T" ] && python tools/rocpd_summary.py stats $T > gpurun_out/s15_kernel_stats.csv
rm -rf gpurun_out/s15_trace
tail -3 gpurun_out/s15_pytest.txt; python - <<'P'
import json
for l in open('gpurun_out/s15_ab.jsonl'):
    if not l.startswith('{'): print(l.strip()[:200]); continue
    d=json.loads(l); print(d['wl'],d['n'],d['variant'],d['ms'],'build',d['build_ms'],'dis',d['discharge_ms'],'rel',d['relabel_ms'],'rb',d['readbacks'],'dl',d['dis_launches'],'rl',d['rel_launches'],'rt',d['rel_tiles'],'same',d['same_labels'])
P
head -6 gpurun_out/s15_kernel_stats.csv | cut -c1-120; tail -2 gpurun_out/s15_trace.log | cut -c1-300
