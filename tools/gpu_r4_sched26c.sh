#!/bin/bash
# after the cross-tile atomics: schedule of the 26-neighbourhood again (markers only; config 3 in the workgroup form)
set -x
mkdir -p gpurun_out
cd /root/repo
V=""
for r in 5 6 8; do for s in 4 5 6; do V="$V rounds_per_relabel=$r,sweeps_sparse26=$s"; done; done
timeout 900 python tools/gpu_ab.py --n 512 --conn 26 --reps 2 base $V max_sweeps=2 max_sweeps=4 grid26_dis=4096 grid26_dis=32768 > gpurun_out/r4_sched26c.jsonl 2>/dev/null
V=""
for r in 2 3 4; do for s in 2 3 4; do V="$V wave_kernels=9,rounds_per_relabel=$r,max_sweeps=$s,sweeps_sparse26=$s"; done; done
timeout 900 python tools/gpu_ab.py --n 512 --conn 26 --regional --reps 3 base wave_kernels=9 $V >> gpurun_out/r4_sched26c.jsonl 2>/dev/null
timeout 900 python tools/gpu_ab.py --n 512 --conn 26 --wl hard --reps 2 base rounds_per_relabel=8 rounds_per_relabel=12 rounds_per_relabel=12,sweeps_sparse26=4 >> gpurun_out/r4_sched26c.jsonl 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/r4_sched26c.jsonl"):
    d = json.loads(l); print(d["n"], d["wl"], d["regional"], d["variant"], d["ms"], d["discharge_ms"], d["relabel_ms"], d["relabels"], d["phases"], d["dis_tiles"], d["rel_tiles"], d["same_labels"])
PY
