# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r6_ab_visit_budget2.jsonl; : > $O
timeout 300 python tools/gpu_ab.py --n 256 --tag budget --reps 9 base visit_budget_radial=5 visit_budget_radial=6 visit_budget_radial=7 >> $O 2>&1
timeout 300 python tools/gpu_ab.py --n 384 --tag budget --reps 5 base visit_budget_radial=4 visit_budget_radial=5 visit_budget_radial=6 >> $O 2>&1
timeout 300 python tools/gpu_ab.py --n 128 --tag budget --reps 9 base visit_budget_radial=4 visit_budget_radial=5 >> $O 2>&1
timeout 300 python tools/gpu_ab.py --n 640 --tag budget --reps 3 base visit_budget_radial=4 visit_budget_radial=5 visit_budget_radial=6 >> $O 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_visit_budget2.jsonl"):
    if not l.startswith("{"): print(l[:200]); continue
    d = json.loads(l)
    print(d["wl"], d["n"], d["variant"], d["ms"], "dis", d["discharge_ms"], "rel", d["relabel_ms"], "tiles", d["dis_tiles"], "phases", d["phases"], "relabels", d["relabels"], d["same_labels"])
PY
