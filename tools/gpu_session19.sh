#!/bin/bash
# round 3, GPU session 19: exact_sink_tiles in its automatic mode, adaptive_rounds 2
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/s19_ab.jsonl; : > $O; rm -f gpurun_out/parity_relaxations.jsonl
( MEDPY_SKIP_BIG_IDS=1 timeout 1000 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/s19_pytest.txt
timeout 300 python tools/gpu_ab.py --n 512 --tag s19 base exact_sink_tiles=0 exact_sink_tiles=2 base >> $O 2>&1
timeout 300 python tools/gpu_ab.py --n 512 --wl hard --reps 2 --tag s19 base exact_sink_tiles=0 >> $O 2>&1
timeout 600 python tools/gpu_ab.py --n 512 --wl ties --reps 2 --tag s19 base exact_sink_tiles=0 max_sweeps=8 max_sweeps=16 >> $O 2>&1
tail -3 gpurun_out/s19_pytest.txt; python - <<'P'
import json
for l in open('gpurun_out/s19_ab.jsonl'):
    if not l.startswith('{'): print(l.strip()[:200]); continue
    d=json.loads(l); print(d['wl'],d['n'],d['variant'],d['ms'],'dis',d['discharge_ms'],'rel',d['relabel_ms'],'relabels',d['relabels'],'phases',d['phases'],'dt',d['dis_tiles'],'rt',d['rel_tiles'],'rb',d['readbacks'],'same',d['same_labels'])
P
