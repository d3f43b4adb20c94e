# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash tools/gpu_evidence.sh r6 slabs 2>&1 | tail -8
timeout 1500 python tools/gpu_slab_scaling.py 256 512 26 1 2 4 8 > gpurun_out/r6_slab_scaling_one_gpu_26conn.jsonl 2>&1
cut -c1-300 gpurun_out/r6_slab_scaling_one_gpu_26conn.jsonl
python -c "
import __graft_entry__ as g
g.smoke()" 2>&1 | tail -4
