#!/bin/bash
# round 3, GPU session 9: list shards 1 vs 16 (6- and 26-neighbourhood)
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/s9_ab.jsonl; : > $O
timeout 400 python tools/gpu_ab.py --n 512 --tag tree base list_shards=1 base list_shards=1 >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 512 --wl hard --tag tree base list_shards=1 >> $O 2>&1
timeout 200 python tools/gpu_ab.py --n 256 --tag tree base list_shards=1 >> $O 2>&1
timeout 300 python tools/gpu_ab.py --n 256 --conn 26 --tag tree base list_shards=1 >> $O 2>&1
( MEDPY_HIP_PARAMS=list_shards=1 MEDPY_SKIP_BIG_IDS=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slabs.py tests/test_gpu_full_neighbourhood.py -m gpu -x -q 2>&1 | tail -5 ) > gpurun_out/s9_pytest.txt
timeout 300 python bench.py --no-cpu --config 3 > gpurun_out/s9_bench_config3.json 2>> gpurun_out/s9_bench.err
MEDPY_HIP_PARAMS=list_shards=1 timeout 300 python bench.py --no-cpu --config 3 > gpurun_out/s9_bench_config3_s1.json 2>> gpurun_out/s9_bench.err
tail -3 gpurun_out/s9_pytest.txt; cut -c1-330 $O; cut -c1-300 gpurun_out/s9_bench_config3.json; echo; cut -c1-300 gpurun_out/s9_bench_config3_s1.json
