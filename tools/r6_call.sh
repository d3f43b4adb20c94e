# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r6_pytest.txt; cat gpurun_out/r6_pytest.txt
timeout 900 python bench.py 2> gpurun_out/r6_bench.err | tail -1 > gpurun_out/r6_bench_n1.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_bench_n1.json"))
print(d["value"], d["ms_per_step"], d["roofline"])
print(json.dumps(d["api_end_to_end"])[:1500])
for k, v in d["also"].items():
    print(k, v["ms_per_step"], v["frac"], v.get("labels_match_reference"), v["validation_all_zero"], v.get("radial_cycles"))
PY
tail -5 gpurun_out/r6_bench.err
