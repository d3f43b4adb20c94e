#!/bin/bash
# round 3, GPU session 16: schedule variants on the tie-heavy and the weak-contrast volume
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/s16_ab.jsonl; : > $O
timeout 600 python tools/gpu_ab.py --n 512 --wl ties --reps 2 --tag s16 base wave_kernels=13 max_sweeps=24 rounds_per_relabel=4 rounds_per_relabel=16 adaptive_rounds=0 adaptive_rounds=1 wave_kernels=13,max_sweeps=24 >> $O 2>&1
timeout 300 python tools/gpu_ab.py --n 512 --wl hard --reps 2 --tag s16 base wave_kernels=13 max_sweeps=24 rounds_per_relabel=16 adaptive_rounds=1 >> $O 2>&1
python - <<'P'
import json
for l in open('gpurun_out/s16_ab.jsonl'):
    if not l.startswith('{'): print(l.strip()[:200]); continue
    d=json.loads(l); print(d['wl'],d['n'],d['variant'],d['ms'],'dis',d['discharge_ms'],'rel',d['relabel_ms'],'relabels',d['relabels'],'phases',d['phases'],'dt',d['dis_tiles'],'rt',d['rel_tiles'],'rb',d['readbacks'],'same',d['same_labels'])
P
