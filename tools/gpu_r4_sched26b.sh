#!/bin/bash
# 26-neighbourhood without a regional term: sweep budget of sparse phases x rounds between relabels (relabels are cheap now)
set -x
mkdir -p gpurun_out
cd /root/repo
V=""
for r in 4 6 8; do for s in 2 3 4 5 6; do V="$V rounds_per_relabel=$r,sweeps_sparse26=$s"; done; done
timeout 900 python tools/gpu_ab.py --n 512 --conn 26 --reps 2 base $V > gpurun_out/r4_sched26b.jsonl 2>/dev/null
timeout 900 python tools/gpu_ab.py --n 512 --conn 26 --wl hard --reps 2 base sweeps_sparse26=3 sweeps_sparse26=4 sweeps_sparse26=5 rounds_per_relabel=8,sweeps_sparse26=4 rounds_per_relabel=12,sweeps_sparse26=4 >> gpurun_out/r4_sched26b.jsonl 2>/dev/null
timeout 900 python tools/gpu_ab.py --n 256 --conn 26 --reps 3 base sweeps_sparse26=3 sweeps_sparse26=4 sweeps_sparse26=5 >> gpurun_out/r4_sched26b.jsonl 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/r4_sched26b.jsonl"):
    d = json.loads(l); print(d["n"], d["wl"], d["variant"], d["ms"], d["discharge_ms"], d["relabel_ms"], d["relabels"], d["phases"], d["dis_tiles"], d["rel_tiles"], d["same_labels"])
PY
