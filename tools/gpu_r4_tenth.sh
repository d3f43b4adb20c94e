#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_cli_io.py tests/test_overlay_reference_script.py -m gpu -x -q > gpurun_out/r4_pytest_d.txt 2>&1; tail -4 gpurun_out/r4_pytest_d.txt
timeout 600 python tools/gpu_workloads.py 512 > gpurun_out/r4_workloads_a.jsonl 2>&1; cut -c1-300 gpurun_out/r4_workloads_a.jsonl
(SLAB_TOTAL_PLANES=2048 timeout 2400 python tools/gpu_slab_scaling.py 256 512 26 1 2 8 > gpurun_out/r4_slab_scaling_26.jsonl 2> gpurun_out/r4_slab_scaling_26.err; tail -3 gpurun_out/r4_slab_scaling_26.err)
python - <<'PY'
import json
for l in open('gpurun_out/r4_slab_scaling_26.jsonl'):
    d=json.loads(l); print(d['slabs'], d['wall_ms'], 'kernel ms/slab', d['kernel_ms_per_slab'], 'relabel passes', d.get('relabel_passes', d.get('relabel_launches')), 'phases', d['phases'], 'exch', d.get('exchanges'), 'red', d.get('reductions'), 'dis tiles', d['discharge_tiles'], 'rel tiles', d['relabel_tiles'], d['labels_sha256'][:12])
PY
