# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_validate.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -15 > gpurun_out/r6_pytest_slabs.txt; cat gpurun_out/r6_pytest_slabs.txt
for N in 8 1; do timeout 1500 python tools/gpu_slab_scaling.py 256 1024 6 $N; done > gpurun_out/r6_slab_scaling_one_gpu.jsonl 2>&1
cut -c1-1200 gpurun_out/r6_slab_scaling_one_gpu.jsonl
