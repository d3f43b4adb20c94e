# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r6_cut_pairs_overflow_path.txt
echo "library built with -DMGC_CUT_PAIRS=16 (mgc_cut_tile_general: layers with more than 16 paying pairs are evaluated in place): the GPU tests that compare cut values" > $O
MEDPY_HIP_LIB=$GRAFT_REPO_ROOT/build/lib_cp16.so timeout 1500 python -m pytest tests/test_gpu_full_neighbourhood.py tests/test_gpu_large.py tests/test_gpu_parity.py tests/test_gpu_validate.py -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | tail -n 3 >> $O
MEDPY_HIP_LIB=$GRAFT_REPO_ROOT/build/lib_cp16.so timeout 300 python tools/gpu_ab.py --lib build/lib_cp16.so --n 512 --conn 26 --regional --tag cp16 --reps 2 base 2>&1 | cut -c1-330 >> $O
timeout 300 python tools/gpu_ab.py --n 512 --conn 26 --regional --tag tree --reps 2 base 2>&1 | cut -c1-330 >> $O
cat $O
