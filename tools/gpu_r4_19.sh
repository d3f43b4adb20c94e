#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python tools/gpu_ab.py --n 512 --conn 6 --regional --reps 3 base rounds_per_relabel=3 rounds_per_relabel=4,max_sweeps=6 exact_sink_tiles=0 exact_sink_tiles=2 > gpurun_out/r4_regional6.jsonl 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r4_regional6.jsonl'):
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    print(d['wl'], d['conn'], d['variant'], d['ms'], 'build', d['build_ms'], 'dis', d['discharge_ms'], 'rel', d['relabel_ms'], 'relabels', d['relabels'], 'phases', d['phases'], 'tiles', d['dis_tiles'], d['rel_tiles'], d['same_labels'])
PY
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt6 -- python $GRAFT_REPO_ROOT/tools/gpu_ab.py --n 512 --conn 6 --regional --reps 2 base > /tmp/kt6.log 2>&1
D=$(find /tmp/kt6 -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tools/rocpd_summary.py stats $D | head -14 | cut -c1-150
