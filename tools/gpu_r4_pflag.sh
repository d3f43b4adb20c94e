#!/bin/bash
# block-form 26-neighbourhood discharge: flagged sparse hand-off + list counter cleared inside the kernel, against the commit before
set -x
mkdir -p gpurun_out
cd /root/repo
timeout 900 python -m pytest tests/test_gpu_full_neighbourhood.py tests/test_gpu_slabs.py -m gpu -x -q -k "not bench_size" > gpurun_out/r4_pflag_parity.txt 2>&1; tail -3 gpurun_out/r4_pflag_parity.txt
: > gpurun_out/r4_pflag.jsonl
for L in "" "--lib build/lib_prev.so --tag prev"; do
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --reps 2 $L base sweeps_sparse26=4 sweeps_sparse26=6 sweeps_sparse26=8 >> gpurun_out/r4_pflag.jsonl 2>/dev/null
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --wl hard --reps 2 $L base >> gpurun_out/r4_pflag.jsonl 2>/dev/null
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --regional --reps 3 $L base prepush=0 wave_kernels=9 >> gpurun_out/r4_pflag.jsonl 2>/dev/null
timeout 600 python tools/gpu_ab.py --n 256 --conn 26 --reps 3 $L base >> gpurun_out/r4_pflag.jsonl 2>/dev/null
done
python - <<'PY'
import json
for l in open("gpurun_out/r4_pflag.jsonl"):
    d = json.loads(l); print(d["tag"], d["n"], d["wl"], d["regional"], d["variant"], d["ms"], d["discharge_ms"], d["relabel_ms"], d["relabels"], d["phases"], d["dis_tiles"], d["rel_tiles"], d["same_labels"])
PY
timeout 300 python tools/gpu_sections26.py 512 0 2>&1 | grep -v Warn | tail -8
