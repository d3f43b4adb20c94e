#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
for V in tree v4; do
  if [ $V = tree ]; then unset MEDPY_HIP_LIB; else export MEDPY_HIP_LIB=$GRAFT_REPO_ROOT/build/lib_$V.so; fi
  echo "== lib $V"
  timeout 300 python tools/gpu_wave_ab.py 512 sphere 1:0:0 9:0:0 2>&1 | tee -a gpurun_out/ab_relv.jsonl
  timeout 300 python tools/gpu_wave_ab.py 512 hard 1:0:0 9:0:0 9:0:12 2>&1 | tee -a gpurun_out/ab_relv.jsonl
done
unset MEDPY_HIP_LIB
timeout 300 python tools/gpu_wave_ab.py 256 sphere 1:0:0 9:0:0 2>&1 | tee -a gpurun_out/ab_relv.jsonl
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error|Error|assert|^E " | head -20 | tee gpurun_out/pytest_i.txt
