# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
rm -f gpurun_out/parity_relaxations.jsonl
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -8 > gpurun_out/r6_pytest.txt; cat gpurun_out/r6_pytest.txt
[ -f gpurun_out/parity_relaxations.jsonl ] && cp gpurun_out/parity_relaxations.jsonl gpurun_out/r6_parity_relaxations.jsonl
( timeout 300 python tools/gpu_ab.py --n 256 --reps 5 --tag policy base; timeout 300 python tools/gpu_ab.py --n 512 --reps 5 --tag policy base; timeout 300 python tools/gpu_ab.py --n 512 --wl hard --reps 3 --tag policy base ) 2>&1 | cut -c1-260
SLAB_TOTAL_PLANES=2048 timeout 900 python tools/gpu_slab_scaling.py 256 1024 6 8 2>&1 | cut -c1-600
