# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/gpu_evidence.sh r6 tests profile config3 bench timeline workloads 2>&1 | tail -40
