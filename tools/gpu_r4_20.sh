#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_validate.py tests/test_gpu_edge_cases.py -m gpu -x -q > gpurun_out/r4_pytest_i.txt 2>&1; tail -4 gpurun_out/r4_pytest_i.txt
timeout 600 python tools/gpu_ab.py --n 512 --conn 6 --regional --reps 3 base prepush=0 rounds_per_relabel=2 rounds_per_relabel=4 rounds_per_relabel=3,max_sweeps=6 rounds_per_relabel=3,exact_sink_tiles=0 > gpurun_out/r4_regional6_b.jsonl 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r4_regional6_b.jsonl'):
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    print(d['wl'], d['conn'], d['variant'], d['ms'], 'build', d['build_ms'], 'dis', d['discharge_ms'], 'rel', d['relabel_ms'], 'relabels', d['relabels'], 'phases', d['phases'], 'tiles', d['dis_tiles'], d['rel_tiles'], d['same_labels'])
PY
timeout 200 python bench.py --steps 5 --no-cpu 2>/dev/null | cut -c1-200
