# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r6_ab_max_sweeps.jsonl; : > $O
timeout 600 python tools/gpu_ab.py --n 512 --tag sweeps --reps 7 base max_sweeps=14 max_sweeps=16 max_sweeps=18 max_sweeps=20 base max_sweeps=14 >> $O 2>&1
timeout 300 python tools/gpu_ab.py --n 256 --tag sweeps --reps 9 base max_sweeps=14 max_sweeps=16 >> $O 2>&1
timeout 300 python tools/gpu_ab.py --n 384 --tag sweeps --reps 5 base max_sweeps=14 max_sweeps=16 >> $O 2>&1
timeout 300 python tools/gpu_ab.py --n 512 --wl hard --tag sweeps --reps 3 base max_sweeps=14 max_sweeps=16 >> $O 2>&1
timeout 300 python tools/gpu_ab.py --n 512 --wl ct --tag sweeps --reps 3 base max_sweeps=14 max_sweeps=16 >> $O 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_max_sweeps.jsonl"):
    if not l.startswith("{"): print(l[:200]); continue
    d = json.loads(l)
    print(d["wl"], d["n"], d["variant"], d["ms"], "dis", d["discharge_ms"], "rel", d["relabel_ms"], "tiles", d["dis_tiles"], "phases", d["phases"], "relabels", d["relabels"], d["same_labels"])
PY
