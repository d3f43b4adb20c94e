#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_validate.py tests/test_gpu_slabs.py -m gpu -x -q -k "not bench_size" > gpurun_out/r4_pytest_g.txt 2>&1; tail -3 gpurun_out/r4_pytest_g.txt
for N in 512 256 128; do timeout 300 python tools/gpu_ab.py --n $N --reps 4 base lazy_planes=0 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l[:200]); continue
    print(d['n'], d['variant'], d['ms'], 'build', d['build_ms'], 'solve', d['solve_ms'], 'dis', d['discharge_ms'], 'rel', d['relabel_ms'], d['phases'], d['dis_tiles'], d['same_labels'])
"; done
