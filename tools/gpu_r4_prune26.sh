#!/bin/bash
# Step mask of the block-form 26-neighbourhood discharge pruned to the directions the excess lasts for: A/B against every admissible direction
set -x
mkdir -p gpurun_out
cd /root/repo
: > gpurun_out/r4_prune26.jsonl
for L in "" "--lib build/lib_alladm.so --tag alladm"; do
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --reps 2 $L base sweeps_sparse26=4 sweeps_sparse26=12 >> gpurun_out/r4_prune26.jsonl 2>/dev/null
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --wl hard --reps 2 $L base >> gpurun_out/r4_prune26.jsonl 2>/dev/null
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --regional --reps 3 $L base >> gpurun_out/r4_prune26.jsonl 2>/dev/null
timeout 600 python tools/gpu_ab.py --n 256 --conn 26 --reps 3 $L base >> gpurun_out/r4_prune26.jsonl 2>/dev/null
done
python - <<'PY'
import json
for l in open("gpurun_out/r4_prune26.jsonl"):
    d = json.loads(l); print(d["tag"], d["n"], d["wl"], d["regional"], d["variant"], d["ms"], d["discharge_ms"], d["relabel_ms"], d["relabels"], d["phases"], d["dis_tiles"], d["rel_tiles"], d["same_labels"])
PY
