# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/gpu_evidence.sh r6 tests bench profile timeline config3 workloads slabs 2>&1 | tail -60
