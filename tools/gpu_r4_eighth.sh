#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_labels.py tests/test_gpu_full_neighbourhood.py -m gpu -x -q > gpurun_out/r4_pytest_b.txt 2>&1; tail -2 gpurun_out/r4_pytest_b.txt
timeout 300 python bench.py --config 3 --steps 5 --warmup 1 --no-cpu 2>/dev/null | cut -c1-260
(timeout 2400 python tools/gpu_slab_scaling.py 256 1024 6 1 2 4 8 > gpurun_out/r4_slab_scaling_6.jsonl 2> gpurun_out/r4_slab_scaling_6.err; tail -3 gpurun_out/r4_slab_scaling_6.err)
python - <<'PY'
import json
for l in open('gpurun_out/r4_slab_scaling_6.jsonl'):
    d=json.loads(l); print(d['slabs'], d['wall_ms'], 'kernel ms/slab', d['kernel_ms_per_slab'], 'relabel passes', d.get('relabel_passes', d.get('relabel_launches')), 'phases', d['phases'], 'exch', d.get('exchanges'), 'red', d.get('reductions'), 'dis tiles', d['discharge_tiles'], 'rel tiles', d['relabel_tiles'], d['labels_sha256'][:12])
PY
