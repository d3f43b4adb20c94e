# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r6_ab_repeat_after_flood_big.jsonl; : > $O
for P in "" "repeat_steps=1" "" "repeat_steps=1"; do
  echo "# SLAB_PARAMS=$P" >> $O
  SLAB_TOTAL_PLANES=2048 SLAB_PARAMS=$P timeout 900 python tools/gpu_slab_scaling.py 256 1024 6 1 >> $O 2>&1
done
for P in "repeat_steps=1" "" "repeat_steps=1"; do
  echo "# SLAB_PARAMS=$P" >> $O
  SLAB_PARAMS=$P timeout 900 python tools/gpu_slab_scaling.py 256 1024 6 8 >> $O 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_repeat_after_flood_big.jsonl"):
    if not l.startswith("{"): print(l[:100].strip()); continue
    d = json.loads(l)
    print(' ', d.get('slabs'), d.get('shape'), [round(x) for x in d.get('kernel_ms_per_slab',[])], 'wall', d.get('wall_ms'), 'dis', d.get('discharge_ms'), 'rel', d.get('relabel_ms'), 'phases', d.get('phases'), 'exch', d.get('exchanges'), str(d.get('labels_sha256'))[:10])
PY
