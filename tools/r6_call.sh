# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( for L in build/lib_before_z.so ""; do
    timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --tag "${L:-tree_one_z_vote}" --reps 7 base
    timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --wl hard --tag "${L:-tree_one_z_vote}" --reps 3 base
    timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 256 --tag "${L:-tree_one_z_vote}" --reps 7 base
    timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --wl ct --tag "${L:-tree_one_z_vote}" --reps 3 base
    timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --wl ties --tag "${L:-tree_one_z_vote}" --reps 1 base
  done ) > gpurun_out/r6_ab_z_votes.jsonl 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_z_votes.jsonl"):
    if not l.startswith("{"): print(l[:200]); continue
    d = json.loads(l)
    print(d["tag"], d["wl"], d["n"], d["ms"], "dis", d["discharge_ms"], "rel", d["relabel_ms"], "tiles", d["dis_tiles"], "phases", d["phases"], "relabels", d["relabels"], d["flow"])
PY
