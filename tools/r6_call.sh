# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT; G=gpurun_out
for V in tree act; do
  OUT=$ROOT/$G/prof_act; rm -rf $OUT; mkdir -p $OUT
  if [ $V = tree ]; then unset MEDPY_HIP_LIB; else export MEDPY_HIP_LIB=$ROOT/build/lib_$V.so; fi
  ( cd /tmp; export TMPDIR=/tmp; timeout 200 rocprofv3 --kernel-trace -d $OUT/tl -- python $ROOT/bench.py --config 3 --steps 2 --warmup 1 --no-cpu --no-extras > $OUT/tl.log 2>&1 )
  D=$(find $OUT/tl -name "*.db" | head -1)
  [ -n "$D" ] && python tools/rocpd_summary.py timeline $D > $G/r6_act_$V.csv
  echo $V; grep "k26_activate_w" $G/r6_act_$V.csv | tail -n 6
done
