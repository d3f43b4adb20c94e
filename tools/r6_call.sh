# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r6_certificate.txt; : > $O
for V in trace=1 trace=1,certify=0; do
  echo "== 512^3 $V" >> $O
  timeout 300 python tools/gpu_ab.py --n 512 --reps 3 $V 2>&1 | grep -E "certificate|^\{|active tiles" | cut -c1-260 >> $O
done
for C in 1 0; do
  echo "== 2048x1024x1024 one handle, certify=$C" >> $O
  MEDPY_HIP_PARAMS="trace=1,certify=$C" SLAB_TOTAL_PLANES=2048 timeout 900 python tools/gpu_slab_scaling.py 256 1024 6 1 2>&1 | grep -E "certificate|^\{|active tiles" | cut -c1-420 >> $O
done
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_validate.py tests/test_gpu_edge_cases.py -m gpu -q -x 2>&1 | grep -E "passed|failed" | tail -2 >> $O
cat $O
