# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT; G=gpurun_out
O=gpurun_out/r6_ab_cut_value26_two_launches.jsonl; : > $O
for L in build/lib_head.so ""; do
  T=${L:-tree}
  timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --conn 26 --regional --tag "$T" --reps 3 base >> $O 2>&1
  timeout 300 python tools/gpu_ab.py ${L:+--lib $L} --n 512 --regional --tag "$T" --reps 3 base >> $O 2>&1
done
OUT=$ROOT/$G/prof_c3tl; mkdir -p $OUT; ( cd /tmp; export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace -d $OUT/tl -- python $ROOT/bench.py --config 3 --steps 1 --warmup 1 --no-cpu --no-extras > $OUT/tl.log 2>&1 )
D=$(find $OUT/tl -name "*.db" | head -1); [ -n "$D" ] && python tools/rocpd_summary.py timeline $D > $G/r6_config3_timeline.csv; rm -rf $OUT/tl
grep -E "k_cut_value26|k_labels8" $G/r6_config3_timeline.csv | tail -n 4
timeout 2400 python -m pytest tests/test_gpu_full_neighbourhood.py tests/test_gpu_large.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_validate.py -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | tail -n 3
python - <<'PY'
import json
for l in open("gpurun_out/r6_ab_cut_value26_two_launches.jsonl"):
    if not l.startswith("{"): print(l[:200]); continue
    d = json.loads(l)
    print(d["tag"], d["wl"], d["n"], d["conn"], d["regional"], d["ms"], "build", d["build_ms"], "solve", d["solve_ms"], "dis", d["discharge_ms"], "rel", d["relabel_ms"], repr(d["flow"]), d["same_labels"], d["dis_tiles"], d["rel_tiles"], d["phases"])
PY
