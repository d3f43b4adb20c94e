#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
: > gpurun_out/ab_occ.jsonl
for V in tree w2 w4; do
  if [ $V = tree ]; then unset MEDPY_HIP_LIB; else export MEDPY_HIP_LIB=$GRAFT_REPO_ROOT/build/lib_$V.so; fi
  echo "== lib $V" | tee -a gpurun_out/ab_occ.jsonl
  timeout 300 python tools/gpu_wave_ab.py 512 sphere 1:0:0 5:0:0 1:8:0 1:16:0 1:0:10 2>&1 | tee -a gpurun_out/ab_occ.jsonl
done
unset MEDPY_HIP_LIB
echo "== 128/256 tree"; timeout 300 python tools/gpu_wave_ab.py 128 sphere 0:0:0 1:0:0 2>&1 | tee gpurun_out/ab128.jsonl
timeout 300 python tools/gpu_wave_ab.py 256 sphere 0:0:0 1:0:0 2>&1 | tee gpurun_out/ab256.jsonl
echo "== pytest" ; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tee gpurun_out/pytest_c.txt
