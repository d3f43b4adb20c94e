#!/bin/bash
# round 4, first GPU call: the 26-neighbourhood wave discharge (k26_discharge_w) -- parity, then A/B against k26_discharge
set -x
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
(/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probes/atomic_f64_probe.hip -o /tmp/atomic_f64_probe && /tmp/atomic_f64_probe) > gpurun_out/r4_atomic_f64_probe.txt 2>&1
MEDPY_HIP_PARAMS=wave_kernels=41 timeout 600 python -m pytest tests/test_gpu_full_neighbourhood.py -m gpu -x -q -k as_shipped > gpurun_out/r4_w26_parity.txt 2>&1
tail -5 gpurun_out/r4_w26_parity.txt
timeout 300 python tools/gpu_ab.py --n 256 --conn 26 --regional --reps 2 base wave_kernels=41 wave_kernels=41,w26_passes=1 wave_kernels=41,w26_flags=1 wave_kernels=41,max_sweeps=4 > gpurun_out/r4_w26_ab256.jsonl 2>&1
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --regional --reps 2 base wave_kernels=41 wave_kernels=41,w26_passes=1 wave_kernels=41,w26_passes=3 wave_kernels=41,w26_flags=1 wave_kernels=41,max_sweeps=4 wave_kernels=41,max_sweeps=5,w26_flags=1 wave_kernels=41,wave_grid26=2048 > gpurun_out/r4_w26_ab512.jsonl 2>&1
timeout 600 python tools/gpu_ab.py --n 512 --conn 26 --reps 2 base wave_kernels=41 wave_kernels=41,w26_flags=1 wave_kernels=41,sweeps_sparse26=12 > gpurun_out/r4_w26_ab512_noreg.jsonl 2>&1
cat gpurun_out/r4_w26_ab256.jsonl gpurun_out/r4_w26_ab512.jsonl gpurun_out/r4_w26_ab512_noreg.jsonl | cut -c1-400
