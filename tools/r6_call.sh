# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r6_ab_watched_supports6.jsonl; : > $O
for R in 1 2; do for L in build/lib_head.so ""; do
  T=${L:-tree}
  timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --tag "$T" --reps 5 base >> $O 2>&1
done; done
for L in build/lib_head.so ""; do
  T=${L:-tree}
  timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --wl hard --tag "$T" --reps 3 base >> $O 2>&1
  timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --wl ct --tag "$T" --reps 3 base >> $O 2>&1
  timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 256 --tag "$T" --reps 5 base >> $O 2>&1
  timeout 400 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --wl ties --tag "$T" --reps 1 base >> $O 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_watched_supports6.jsonl"):
    if not l.startswith("{"): print(l[:200]); continue
    d = json.loads(l)
    print(d["tag"], d["wl"], d["n"], d["ms"], "solve", d["solve_ms"], "dis", d["discharge_ms"], "rel", d["relabel_ms"], repr(d["flow"]), d["dis_tiles"], d["rel_tiles"], d["phases"], d["relabels"])
PY
