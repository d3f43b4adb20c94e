#!/bin/bash
# 26-neighbourhood without a regional term: schedule sweep now that incremental relabels are cheap
set -x
mkdir -p gpurun_out
cd /root/repo
V=""
for r in 3 4 6 8 12; do for s in 2 3 4 6; do V="$V rounds_per_relabel=$r,max_sweeps=$s"; done; done
timeout 900 python tools/gpu_ab.py --n 512 --conn 26 --reps 2 base $V adaptive_rounds=0 adaptive_rounds=3 adaptive_rounds=20 relabel_batch=4 relabel_batch=16 > gpurun_out/r4_sched26.jsonl 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/r4_sched26.jsonl"):
    d = json.loads(l); print(d["variant"], d["ms"], d["discharge_ms"], d["relabel_ms"], d["relabels"], d["phases"], d["dis_tiles"], d["rel_tiles"], d["same_labels"])
PY
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --wl hard --reps 2 base rounds_per_relabel=4 rounds_per_relabel=8,max_sweeps=4 rounds_per_relabel=12 > gpurun_out/r4_sched26_hard.jsonl 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/r4_sched26_hard.jsonl"):
    d = json.loads(l); print("hard", d["variant"], d["ms"], d["discharge_ms"], d["relabel_ms"], d["relabels"], d["phases"], d["dis_tiles"], d["rel_tiles"], d["same_labels"])
PY
