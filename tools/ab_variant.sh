#!/bin/bash
# A/B helper for kernel experiments (development aid), economical with GPU minutes:
#   HERE (no GPU):  bash tools/ab_variant.sh build <name> <patch-file | -D flags...>
#                   -> build/lib_<name>.so from a scratch copy of medpy_amd/csrc with the patch / the -D flags applied;
#                      the tree itself is not modified.  build/ is git-ignored but travels to the GPU box.
#   ON THE GPU BOX: gpurun -- 'bash tools/ab_variant.sh run <name>'
#                   -> parity tests (timeout 300 s: a variant that hangs costs 5 minutes, not the round) and bench.py for the
#                      variant (MEDPY_HIP_LIB) and for the in-tree library, one line each in gpurun_out/ab_<name>.txt
set -u
MODE=${1:?build|run}; NAME=${2:?variant name}; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
case $MODE in
build)
    W=$ROOT/build/ab_$NAME; rm -rf $W; mkdir -p $W/medpy_amd $W/include
    cp -r $ROOT/medpy_amd/csrc $W/medpy_amd/; cp $ROOT/include/medpy_hip.h $W/include/
    FLAGS=()
    for a in "$@"; do
        if [ -f "$a" ]; then P=$(readlink -f "$a"); (cd $W && grep -v '^#' "$P" | git apply --include='medpy_amd/csrc/*' -) || exit 1; else FLAGS+=("$a"); fi
    done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -ldl "${FLAGS[@]}" \
        $W/medpy_amd/csrc/mgc_kernels.hip $W/medpy_amd/csrc/msg_sparse.hip -o $ROOT/build/lib_$NAME.so || exit 1
    rm -rf $W; ls -la $ROOT/build/lib_$NAME.so ;;
run)
    OUT=$ROOT/gpurun_out/ab_$NAME.txt; : > $OUT; cd $ROOT
    for V in variant tree; do
        if [ $V = variant ]; then export MEDPY_HIP_LIB=$ROOT/build/lib_$NAME.so; else unset MEDPY_HIP_LIB; fi
        T=$(timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_slabs.py tests/test_gpu_edge_cases.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -1)
        [ -z "$T" ] && T="TESTS TIMED OUT OR CRASHED"
        B=$(timeout 200 python bench.py --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['phases_ms']['discharge_kernels'], d['phases_ms']['relabel_kernels'])" 2>/dev/null)
        echo "$V: tests [$T] bench ms/step, discharge, relabel: [${B:-NO BENCH LINE}]" >> $OUT
        if [ $V = variant ] && ! echo "$T" | grep -q passed; then echo "variant failed its tests: skipping nothing else, tree run follows" >> $OUT; fi
    done
    cat $OUT ;;
esac
