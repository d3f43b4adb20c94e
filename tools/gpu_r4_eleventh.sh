#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
show() { python - "$1" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    print(d['wl'], d['conn'], d['variant'], d['ms'], 'dis', d['discharge_ms'], 'rel', d['relabel_ms'], 'relabels', d['relabels'], 'phases', d['phases'], 'tiles', d['dis_tiles'], d['rel_tiles'], d['same_labels'])
PY
}
timeout 1200 python tools/gpu_ab.py --n 512 --conn 26 --reps 2 base rounds_per_relabel=4 rounds_per_relabel=8 rounds_per_relabel=12 sweeps_sparse26=3 sweeps_sparse26=5 sweeps_sparse26=12 adaptive_rounds=3 adaptive_rounds=30 rounds_per_relabel=8,sweeps_sparse26=5 rounds_per_relabel=12,sweeps_sparse26=12 incremental_relabel=0 > gpurun_out/r4_sweep26_noreg.jsonl 2>&1; show gpurun_out/r4_sweep26_noreg.jsonl
timeout 1200 python tools/gpu_ab.py --n 512 --wl ties --reps 2 base rounds_per_relabel=12 rounds_per_relabel=16 sink_sweeps=6 sink_sweeps=12 adaptive_rounds=1 exact_sink_tiles=2 incremental_relabel=0 rounds_per_relabel=16,incremental_relabel=0 > gpurun_out/r4_sweep_ties.jsonl 2>&1; show gpurun_out/r4_sweep_ties.jsonl
timeout 600 python tools/gpu_ab.py --n 256 --reps 5 base wave_min_tiles=256 wave_min_tiles=1024 wave_min_tiles=128 rounds_per_relabel=6 rounds_per_relabel=12 check_rounds=8 relabel_batch=16 > gpurun_out/r4_sweep_256.jsonl 2>&1; show gpurun_out/r4_sweep_256.jsonl
