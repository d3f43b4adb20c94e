#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_validate.py -m gpu -x -q -k "not bench_size" > gpurun_out/r4_pytest_c.txt 2>&1; tail -2 gpurun_out/r4_pytest_c.txt
(timeout 2400 python tools/gpu_slab_scaling.py 256 1024 6 2 4 8 > gpurun_out/r4_slab_scaling_6b.jsonl 2> gpurun_out/r4_slab_scaling_6b.err; tail -3 gpurun_out/r4_slab_scaling_6b.err)
python - <<'PY'
import json
for l in open('gpurun_out/r4_slab_scaling_6b.jsonl'):
    d=json.loads(l); print(d['slabs'], d['wall_ms'], 'kernel ms/slab', d['kernel_ms_per_slab'], 'relabel passes', d.get('relabel_passes', d.get('relabel_launches')), 'phases', d['phases'], 'exch', d.get('exchanges'), 'red', d.get('reductions'), 'dis tiles', d['discharge_tiles'], 'rel tiles', d['relabel_tiles'], d['labels_sha256'][:12])
PY
