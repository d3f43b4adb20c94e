#!/bin/bash
cd $GRAFT_REPO_ROOT
for n in 64 128 256 512; do timeout 300 python tools/gpu_wave_ab.py $n sphere 9:0:0 2>&1 | cut -c1-250; done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert|^E " | head
