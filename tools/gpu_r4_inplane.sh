#!/bin/bash
# k_discharge_w: extra rounds of the in-plane directions per sweep; k26_discharge_v (two voxels per thread) now that it has no scratch
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sphere or prepush or slab" 2>&1 | tail -2
: > gpurun_out/r4_inplane.jsonl
timeout 600 python tools/gpu_ab.py --n 512 --reps 4 base inplane_extra=1 inplane_extra=2 inplane_extra=3 inplane_extra=1,max_sweeps=8 inplane_extra=2,max_sweeps=8 inplane_extra=3,max_sweeps=6 >> gpurun_out/r4_inplane.jsonl 2>/dev/null
timeout 600 python tools/gpu_ab.py --n 512 --wl hard --reps 3 base inplane_extra=1 inplane_extra=2 inplane_extra=3 >> gpurun_out/r4_inplane.jsonl 2>/dev/null
timeout 600 python tools/gpu_ab.py --n 512 --wl ties --reps 2 base inplane_extra=2 >> gpurun_out/r4_inplane.jsonl 2>/dev/null
timeout 600 python tools/gpu_ab.py --n 512 --regional --reps 3 base inplane_extra=2 >> gpurun_out/r4_inplane.jsonl 2>/dev/null
timeout 600 python tools/gpu_ab.py --n 256 --reps 5 base inplane_extra=1 inplane_extra=2 inplane_extra=3 >> gpurun_out/r4_inplane.jsonl 2>/dev/null
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --reps 2 base wave_kernels=25 >> gpurun_out/r4_inplane.jsonl 2>/dev/null
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --wl hard --reps 2 base wave_kernels=25 >> gpurun_out/r4_inplane.jsonl 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/r4_inplane.jsonl"):
    d = json.loads(l); print(d["n"], d["conn"], d["wl"], d["regional"], d["variant"], d["ms"], d["discharge_ms"], d["relabel_ms"], d["relabels"], d["phases"], d["dis_tiles"], d["rel_tiles"], d["same_labels"])
PY
