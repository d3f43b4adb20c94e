# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r6_ab_repeat_after_flood_threshold.jsonl; : > $O
timeout 600 python tools/gpu_ab.py --n 512 --tag s512 --reps 5 repeat_steps=1 repeat_flood_min_tiles=0 repeat_flood_min_tiles=2048 repeat_flood_min_tiles=4096 repeat_flood_min_tiles=0 repeat_steps=1 >> $O 2>&1
timeout 600 python tools/gpu_ab.py --n 256 --tag s256 --reps 7 repeat_steps=1 repeat_flood_min_tiles=2048 repeat_flood_min_tiles=4096 >> $O 2>&1
timeout 600 python tools/gpu_ab.py --n 384 --tag s384 --reps 5 repeat_steps=1 repeat_flood_min_tiles=0 repeat_flood_min_tiles=2048 repeat_flood_min_tiles=4096 repeat_steps=1 >> $O 2>&1
timeout 600 python tools/gpu_ab.py --n 640 --tag s640 --reps 3 repeat_steps=1 repeat_flood_min_tiles=0 repeat_flood_min_tiles=4096 >> $O 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_repeat_after_flood_threshold.jsonl"):
    if not l.startswith("{"): print(l[:200]); continue
    d = json.loads(l)
    print(d["tag"], d["n"], d["variant"], d["ms"], "solve", d["solve_ms"], "dis", d["discharge_ms"], "rel", d["relabel_ms"], d["same_labels"], d["dis_tiles"], d["rel_tiles"], d["phases"], d["relabels"])
PY
