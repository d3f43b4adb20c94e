# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT; G=gpurun_out
OUT=$ROOT/$G/prof_c3tl; mkdir -p $OUT; ( cd /tmp; export TMPDIR=/tmp
  timeout 300 rocprofv3 --kernel-trace -d $OUT/tl -- python $ROOT/bench.py --config 3 --steps 1 --warmup 1 --no-cpu --no-extras > $OUT/tl.log 2>&1 )
D=$(find $OUT/tl -name "*.db" | head -1); [ -n "$D" ] && python tools/rocpd_summary.py timeline $D > $G/r6_config3_timeline.csv; rm -rf $OUT/tl
tail -3 $OUT/tl.log | cut -c1-300
timeout 300 python tools/gpu_ab.py --n 512 --conn 26 --regional --tag tree --reps 2 base trace=1 > $G/r6_c3_trace.txt 2>&1
grep -c . $G/r6_config3_timeline.csv
