#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for P in "prepush=1" "prepush=0" "prepush=0,use_filters=1"; do
  echo "== $P" >> gpurun_out/r4_bisect.txt
  MEDPY_HIP_PARAMS=$P timeout 300 python -m pytest "tests/test_gpu_validate.py::test_slabs_validate_globally" -m gpu -x -q 2>&1 | grep -E "passed|failed|violated" >> gpurun_out/r4_bisect.txt
done
cat gpurun_out/r4_bisect.txt
timeout 900 python tools/gpu_ab.py --n 512 --conn 26 --regional --reps 2 base rounds_per_relabel=1 rounds_per_relabel=2 rounds_per_relabel=3 rounds_per_relabel=2,sweeps_sparse26=3 rounds_per_relabel=2,sweeps_sparse26=5 rounds_per_relabel=2,sweeps_sparse26=12 rounds_per_relabel=1,sweeps_sparse26=12 rounds_per_relabel=2,adaptive_rounds=0 rounds_per_relabel=2,grid26_dis=512 > gpurun_out/r4_prepush_sched.jsonl 2>&1
cut -c100-420 gpurun_out/r4_prepush_sched.jsonl
