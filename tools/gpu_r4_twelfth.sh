#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "integer_valued or ties_volume" > gpurun_out/r4_pytest_e.txt 2>&1; tail -4 gpurun_out/r4_pytest_e.txt
bash tools/gpu_r4_eleventh.sh
