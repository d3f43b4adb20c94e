#!/bin/bash
# round 3, GPU session 11: in-library slab driver + bounded single-transfer exchanges (mock RCCL, loopback, gloo), config 3 with / without per-launch events
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
( MEDPY_SKIP_BIG_IDS=1 timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_validate.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/s11_pytest.txt
timeout 300 python bench.py --no-cpu --config 3 > gpurun_out/s11_bench_config3.json 2> gpurun_out/s11_bench.err
MEDPY_HIP_PARAMS=kernel_timing=0 timeout 300 python bench.py --no-cpu --config 3 > gpurun_out/s11_bench_config3_notiming.json 2>> gpurun_out/s11_bench.err
tail -4 gpurun_out/s11_pytest.txt; cut -c1-260 gpurun_out/s11_bench_config3.json; echo; cut -c1-260 gpurun_out/s11_bench_config3_notiming.json
