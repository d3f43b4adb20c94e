#!/bin/bash
# round 3, GPU session 22: an incremental relabel seeds only the reset tiles a label can reach in the first pass
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/s22_ab.jsonl; : > $O
( MEDPY_SKIP_BIG_IDS=1 timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/s22_pytest.txt
timeout 200 python tools/gpu_ab.py --n 512 --tag s22 base base >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 512 --wl hard --reps 2 --tag s22 base >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 512 --wl ties --reps 2 --tag s22 base >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 256 --tag s22 base >> $O 2>&1
tail -2 gpurun_out/s22_pytest.txt; python - <<'P'
import json
for l in open('gpurun_out/s22_ab.jsonl'):
    if not l.startswith('{'): print(l.strip()[:200]); continue
    d=json.loads(l); print(d['wl'],d['n'],d['variant'],d['ms'],'dis',d['discharge_ms'],'rel',d['relabel_ms'],'relabels',d['relabels'],'phases',d['phases'],'dt',d['dis_tiles'],'rt',d['rel_tiles'],'rl',d['rel_launches'],'rb',d['readbacks'])
P
