#!/bin/bash
# ONE script for every GPU evidence run (replaces the per-call session scripts of rounds 2 - 4).  Run on the GPU box through
# gpurun from the repo root; everything lands under gpurun_out/ with the round tag in the name, copy what is to be judged
# into profiles/.
#
#   bash tools/gpu_evidence.sh <tag> <what> [<what> ...]
#
# <tag>   prefix of the output files (r5, r5b ...)
# <what>  tests        the whole GPU test tier (pytest -m gpu), tail of the report + the parity-relaxation records
#         profile      rocprofv3 passes of bench.py (headline): kernel trace + stats, FETCH_SIZE / WRITE_SIZE in passes of
#                      their own, instruction-mix counters (tools/profile_round.sh); pmc_discharge.json
#         timeline     every dispatch of one bench step in order with the gap before it (rocpd_summary.py timeline)
#         bench        the bench line (headline, CPU baseline in the run, config 2 / 3 and the API path inside it)
#         config3      bench.py --config 3 alone + its kernel trace and FETCH / WRITE passes (pmc_discharge26.json)
#         workloads    default schedule on sphere / hard / ties (6-neighbourhood) and the 26-neighbourhood family
#         sections     cycles per section of a discharge visit (needs build/lib_prof.so: tools/ab_variant.sh build prof -DMGCW_PROFILE)
#         slabs        tools/gpu_slab_scaling.py 256 1024 6 {1,2,4,8} on one GPU (time-multiplexed slabs)
#         ab:<args>    tools/gpu_ab.py <args with , for spaces between variants kept as is>, e.g. "ab:--n 512 base first_relabel_radial=0"
# Counters never share a run with traces other than --kernel-trace (the pool's rule for rocprofv3).
set -u
TAG=${1:?tag}; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}; cd $ROOT; mkdir -p gpurun_out
G=gpurun_out
for WHAT in "$@"; do
  case "$WHAT" in
  tests)
    rm -f $G/parity_relaxations.jsonl
    ( timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 ) > $G/${TAG}_pytest.txt
    [ -f $G/parity_relaxations.jsonl ] && cp $G/parity_relaxations.jsonl $G/${TAG}_parity_relaxations.jsonl
    tail -2 $G/${TAG}_pytest.txt ;;
  profile)
    bash tools/profile_round.sh $TAG
    cp $G/pmc_discharge.json profiles/ 2>/dev/null ;;
  timeline)
    OUT=$ROOT/$G/prof_$TAG; mkdir -p $OUT; ( cd /tmp; export TMPDIR=/tmp
      timeout 300 rocprofv3 --kernel-trace -d $OUT/tl -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu --no-extras > $OUT/tl.log 2>&1 )
    D=$(find $OUT/tl -name "*.db" | head -1); [ -n "$D" ] && python tools/rocpd_summary.py timeline $D > $G/${TAG}_timeline.csv; rm -rf $OUT/tl
    python - <<PY
import csv
rows = list(csv.DictReader(open("$G/${TAG}_timeline.csv")))
# the last bench step = the dispatches after the last k_build
last = max(i for i, r in enumerate(rows) if r["kernel"].startswith("k_build"))
step = rows[last:]
for i, r in enumerate(step[1:], 1):  # the step ends where the host takes over (validation, read-out to the host: a gap of milliseconds)
    if float(r["gap_us"]) > 5000.0:
        step = step[:i]
        break
gaps = [float(r["gap_us"]) for r in step[1:]]
print("dispatches in the step: %d; gaps > 5 us: %d, > 20 us: %d; idle between dispatches %.2f ms" % (len(step), sum(g > 5 for g in gaps), sum(g > 20 for g in gaps), sum(g for g in gaps if g > 0) / 1e3))
PY
    ;;
  bench)
    python bench.py 2> $G/${TAG}_bench.err | tail -1 > $G/${TAG}_bench_n1.json; cut -c1-900 $G/${TAG}_bench_n1.json ;;
  config3)
    OUT=$ROOT/$G/prof26; rm -rf $OUT; mkdir -p $OUT; ( cd /tmp; export TMPDIR=/tmp
      timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $ROOT/bench.py --config 3 --no-cpu --no-extras --steps 2 --warmup 1 > $OUT/trace.log 2>&1
      timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -- python $ROOT/bench.py --config 3 --no-cpu --no-extras --steps 1 --warmup 0 > $OUT/fetch.log 2>&1
      timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -- python $ROOT/bench.py --config 3 --no-cpu --no-extras --steps 1 --warmup 0 > $OUT/write.log 2>&1 )
    T=$(find $OUT/trace -name "*.db" | head -1); F=$(find $OUT/fetch -name "*.db" | head -1); W=$(find $OUT/write -name "*.db" | head -1)
    [ -n "$T" ] && python tools/rocpd_summary.py stats $T > $G/${TAG}_config3_trace.csv
    [ -n "$F" ] && python tools/rocpd_summary.py pmc $F > $G/${TAG}_config3_fetch.csv
    [ -n "$W" ] && python tools/rocpd_summary.py pmc $W > $G/${TAG}_config3_write.csv
    [ -n "$F" ] && [ -n "$W" ] && python tools/rocpd_summary.py json $F $W k26_discharge_w > $G/pmc_discharge26.json && cp $G/pmc_discharge26.json profiles/
    rm -rf $OUT
    python bench.py --config 3 --no-cpu --no-extras 2>> $G/${TAG}_bench.err | tail -1 > $G/${TAG}_bench_config3.json; cut -c1-400 $G/${TAG}_bench_config3.json ;;
  workloads)
    timeout 400 python tools/gpu_workloads.py 512 > $G/${TAG}_workloads.jsonl 2>&1
    ( timeout 300 python tools/gpu_ab.py --n 512 --conn 26 --reps 2 base; timeout 300 python tools/gpu_ab.py --n 512 --conn 26 --regional --reps 2 base prepush=0
      timeout 300 python tools/gpu_ab.py --n 512 --conn 26 --wl hard --reps 2 base; timeout 300 python tools/gpu_ab.py --n 512 --regional --reps 3 base prepush=0 ) > $G/${TAG}_workloads26.jsonl 2>&1
    cut -c1-260 $G/${TAG}_workloads.jsonl $G/${TAG}_workloads26.jsonl ;;
  sections)
    MEDPY_HIP_LIB=$ROOT/build/lib_prof.so timeout 200 python tools/gpu_sections.py 512 2>&1 | grep -v Warn | tail -14 > $G/${TAG}_discharge_sections.txt
    MEDPY_HIP_LIB=$ROOT/build/lib_prof.so timeout 200 python tools/gpu_sections26.py 512 0 2>&1 | grep -v Warn | tail -6 > $G/${TAG}_discharge26_sections_noreg.txt
    cat $G/${TAG}_discharge_sections.txt ;;
  slabs)
    timeout 3000 python tools/gpu_slab_scaling.py 256 1024 6 1 2 4 8 > $G/${TAG}_slab_scaling_one_gpu.jsonl 2>&1
    cut -c1-400 $G/${TAG}_slab_scaling_one_gpu.jsonl ;;
  ab:*)
    N=$(ls $G/${TAG}_ab_*.jsonl 2>/dev/null | wc -l)
    timeout 900 python tools/gpu_ab.py ${WHAT#ab:} > $G/${TAG}_ab_$N.jsonl 2>&1; cut -c1-420 $G/${TAG}_ab_$N.jsonl ;;
  *) echo "unknown: $WHAT" ;;
  esac
done
