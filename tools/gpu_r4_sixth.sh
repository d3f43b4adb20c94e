#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_full_neighbourhood.py tests/test_gpu_validate.py tests/test_gpu_large.py tests/test_gpu_slabs.py -m gpu -x -q > gpurun_out/r4_prepush_tests.txt 2>&1
tail -3 gpurun_out/r4_prepush_tests.txt
V=""
for R in 2 3 4 5; do for S in 3 5 8; do V="$V rounds_per_relabel=$R,sweeps_sparse26=$S"; done; done
timeout 900 python tools/gpu_ab.py --n 512 --conn 26 --regional --reps 2 base $V rounds_per_relabel=3,adaptive_rounds=0 rounds_per_relabel=3,max_sweeps=2 rounds_per_relabel=3,sweeps_sparse26=8,grid26_dis=1024 > gpurun_out/r4_prepush_sched2.jsonl 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r4_prepush_sched2.jsonl'):
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    print(d['variant'], d['ms'], 'dis', d['discharge_ms'], 'rel', d['relabel_ms'], 'relabels', d['relabels'], 'phases', d['phases'], 'tiles', d['dis_tiles'], d['same_labels'])
PY
