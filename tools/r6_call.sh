# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r6_ab_timing_stride26.jsonl; : > $O
timeout 900 python tools/gpu_ab.py --n 512 --conn 26 --regional --tag c3 --reps 5 base timing_stride=7 timing_stride=3 kernel_timing=0 base >> $O 2>&1
timeout 900 python tools/gpu_ab.py --n 512 --conn 26 --tag m26 --reps 2 base timing_stride=7 kernel_timing=0 >> $O 2>&1
timeout 900 python tools/gpu_ab.py --n 256 --conn 26 --regional --tag c3_256 --reps 5 base timing_stride=7 kernel_timing=0 >> $O 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_timing_stride26.jsonl"):
    if not l.startswith("{"): print(l[:200]); continue
    d = json.loads(l)
    print(d["tag"], d["n"], d["variant"], d["ms"], "solve", d["solve_ms"], "dis", d["discharge_ms"], "rel", d["relabel_ms"], d["same_labels"], d["dis_tiles"], d["rel_tiles"], d["phases"], d["relabels"])
PY
