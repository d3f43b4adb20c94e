#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
for W in 9 25; do echo "== MEDPY_HIP_WAVE=$W"; MEDPY_HIP_WAVE=$W timeout 300 python tools/gpu_probe26.py 512 2>&1 | cut -c1-420 | tee -a gpurun_out/r2_config3_ab.jsonl; done
timeout 600 python -m pytest tests/test_gpu_full_neighbourhood.py tests/test_gpu_slabs.py tests/test_gpu_validate.py tests/test_gpu_large.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert|^E " | head -20
MEDPY_HIP_WAVE=25 timeout 600 python -m pytest tests/test_gpu_full_neighbourhood.py tests/test_gpu_slabs.py tests/test_gpu_validate.py tests/test_gpu_large.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert|^E " | head -20
