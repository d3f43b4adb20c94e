#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    print(d['wl'], d['conn'], d['variant'], d['ms'], 'build', d['build_ms'], 'dis', d['discharge_ms'], 'rel', d['relabel_ms'], 'relabels', d['relabels'], 'phases', d['phases'], 'tiles', d['dis_tiles'], d['rel_tiles'], d['same_labels'])
PY
}
timeout 900 python tools/gpu_ab.py --n 512 --conn 26 --regional --reps 2 base wave_kernels=41 wave_kernels=41,w26_passes=1 wave_kernels=41,w26_flags=1 wave_kernels=41,max_sweeps=2 wave_kernels=41,max_sweeps=5 rounds_per_relabel=3,sweeps_sparse26=2 rounds_per_relabel=3,sweeps_sparse26=3,max_sweeps=2 rounds_per_relabel=3,sweeps_sparse26=4,max_sweeps=4 > gpurun_out/r4_w26_prepush.jsonl 2>&1; show gpurun_out/r4_w26_prepush.jsonl
timeout 600 python tools/gpu_ab.py --n 256 --conn 26 --regional --reps 3 base prepush=0 > gpurun_out/r4_c3_256.jsonl 2>&1; show gpurun_out/r4_c3_256.jsonl
