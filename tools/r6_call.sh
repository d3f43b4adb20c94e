# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( for L in "" build/lib_vote4.so build/lib_runahead.so build/lib_runahead_pf.so; do
    timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --tag "${L:-tree}" --reps 5 base
    timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --wl hard --tag "${L:-tree}" --reps 3 base
    timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 256 --tag "${L:-tree}" --reps 5 base
  done ) > gpurun_out/r6_ab_votes_runahead.jsonl 2>&1
cut -c1-330 gpurun_out/r6_ab_votes_runahead.jsonl
SLAB_TOTAL_PLANES=2048 timeout 900 python tools/gpu_slab_scaling.py 256 1024 6 8 > gpurun_out/r6_slab_n8_defaults.jsonl 2>&1; cut -c1-700 gpurun_out/r6_slab_n8_defaults.jsonl
rm -f gpurun_out/parity_relaxations.jsonl
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -8 > gpurun_out/r6_pytest.txt; cat gpurun_out/r6_pytest.txt
[ -f gpurun_out/parity_relaxations.jsonl ] && cp gpurun_out/parity_relaxations.jsonl gpurun_out/r6_parity_relaxations.jsonl
