#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_full_neighbourhood.py tests/test_gpu_large.py -m gpu -x -q -k "not 2_31" > gpurun_out/r4_pytest_h.txt 2>&1; tail -2 gpurun_out/r4_pytest_h.txt
MEDPY_HIP_LIB=build/lib_prof26.so timeout 300 python tools/gpu_sections26w.py 512 regional wave_kernels=41,w26_passes=1 > gpurun_out/r4_w26_sections_prepush.txt 2>&1; cat gpurun_out/r4_w26_sections_prepush.txt
timeout 300 python bench.py --config 3 --no-cpu 2>/dev/null | cut -c1-330
