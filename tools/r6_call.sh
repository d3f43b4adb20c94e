# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r6_ab_activation_as_built.jsonl; : > $O
for L in build/lib_head2.so "" build/lib_head2.so ""; do
  T=${L:-tree}
  timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --conn 26 --regional --tag "$T" --reps 4 base >> $O 2>&1
done
for L in build/lib_head2.so ""; do
  T=${L:-tree}
  timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --conn 26 --tag "$T" --reps 2 base >> $O 2>&1
  timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 256 --conn 26 --regional --tag "$T" --reps 4 base >> $O 2>&1
done
timeout 1500 python -m pytest tests/test_gpu_full_neighbourhood.py tests/test_gpu_large.py tests/test_gpu_slabs.py tests/test_gpu_validate.py -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | tail -n 3
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_activation_as_built.jsonl"):
    if not l.startswith("{"): print(l[:200]); continue
    d = json.loads(l)
    print(d["tag"], d["wl"], d["n"], d["conn"], d["regional"], d["ms"], "build", d["build_ms"], "solve", d["solve_ms"], "dis", d["discharge_ms"], "rel", d["relabel_ms"], repr(d["flow"]), d["same_labels"], d["dis_tiles"], d["rel_tiles"], d["phases"])
PY
