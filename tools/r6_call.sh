# one GPU call of round 6 (development aid; edited per call)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( for WL in sphere hard ties ct; do timeout 600 python tools/gpu_ab.py --n 512 --wl $WL --reps 2 --tag radial_on_off radial=0 radial=2 radial=1; done ) > gpurun_out/r6_radial_on_off.jsonl 2>&1
cut -c1-300 gpurun_out/r6_radial_on_off.jsonl
bash tools/gpu_evidence.sh r6 bench profile timeline config3 2>&1 | tail -30
