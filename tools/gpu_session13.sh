#!/bin/bash
# round 3, GPU session 13: schedule knobs (stop_below, check_rounds), relabel with 4 voxels per thread, config 3 after the fused wake-ups
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/s13_ab.jsonl; : > $O
( MEDPY_SKIP_BIG_IDS=1 timeout 600 python -m pytest tests/test_gpu_full_neighbourhood.py tests/test_gpu_slabs.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/s13_pytest.txt
timeout 400 python tools/gpu_ab.py --n 512 --tag tree base stop_below=256 stop_below=1024 check_rounds=2,stop_below=512 check_rounds=8 wave_min_tiles=256 wave_min_tiles=1024 >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 512 --tag v4 --lib $PWD/build/lib_v4.so base >> $O 2>&1
timeout 300 python tools/gpu_ab.py --n 512 --wl hard --tag tree base stop_below=1024 >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 512 --wl hard --tag v4 --lib $PWD/build/lib_v4.so base >> $O 2>&1
timeout 300 python bench.py --no-cpu --config 3 > gpurun_out/s13_bench_config3.json 2> gpurun_out/s13_bench.err
tail -3 gpurun_out/s13_pytest.txt; python - <<'P'
import json
for l in open('gpurun_out/s13_ab.jsonl'):
    if not l.startswith('{'): print(l.strip()[:200]); continue
    d=json.loads(l); print(d['tag'],d['wl'],d['n'],d['variant'],d['ms'],'dis',d['discharge_ms'],'rel',d['relabel_ms'],'rb',d['readbacks'],'dl',d['dis_launches'],'rl',d['rel_launches'],'relabels',d['relabels'],'dt',d['dis_tiles'],'same',d['same_labels'])
P
cut -c1-260 gpurun_out/s13_bench_config3.json
