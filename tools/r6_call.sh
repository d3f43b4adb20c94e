# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r6_resident_waves_curve.jsonl; : > $O
timeout 300 python tools/gpu_ab.py --n 512 --tag w2 --reps 3 wave_grid_dis=256 wave_grid_dis=512 wave_grid_dis=768 wave_grid_dis=1024 wave_grid_dis=1536 wave_grid_dis=2048 >> $O 2>&1
timeout 300 python tools/gpu_ab.py --lib build/lib_w3.so --n 512 --tag w3 --reps 3 wave_grid_dis=2048 wave_grid_dis=2560 wave_grid_dis=3072 >> $O 2>&1
python - <<'PY'
import json
for l in open("gpurun_out/r6_resident_waves_curve.jsonl"):
    if not l.startswith("{"): print(l[:200]); continue
    d = json.loads(l)
    print(d["tag"], d["variant"], d["ms"], "dis", d["discharge_ms"], "tiles", d["dis_tiles"], "visits/us", round(d["dis_tiles"]/d["discharge_ms"]/1e3,1))
PY
