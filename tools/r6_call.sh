# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
O=gpurun_out/r6_ab_stream_loads.jsonl; : > $O
for R in 1 2; do for L in "" build/lib_sld.so; do
  T=${L:-tree}
  timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --tag "$T" --reps 7 base >> $O 2>&1
  timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --wl hard --tag "$T" --reps 3 base >> $O 2>&1
done; done
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_stream_loads.jsonl"):
    if not l.startswith("{"): print(l[:200]); continue
    d = json.loads(l)
    print(d["tag"], d["wl"], d["n"], d["ms"], "dis", d["discharge_ms"], "rel", d["relabel_ms"], d["same_labels"])
PY
