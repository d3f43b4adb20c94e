#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_validate.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert|^E " | head
timeout 600 python tools/gpu_r2_o.py 512 1024 2>&1 | grep "slab" | cut -c1-400
