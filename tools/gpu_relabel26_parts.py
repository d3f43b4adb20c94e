"""What a pass of the 26-neighbourhood global relabel is made of (development aid): config 3's graph, ONE outer iteration
(max_outer=1: the first relabel and its colour rounds), meant to run under `rocprofv3 --kernel-trace` with an experiment
library (MEDPY_HIP_LIB) that leaves a part of mgc26_relabel_tile out; only the durations of k26_relabel_all / the first
k26_relabel_list are read, the labels of such a library mean nothing."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd import synthetic  # noqa: E402
from medpy_amd.graphcut.graph import VoxelGraph  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
regional = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
s = synthetic.sphere((n, n, n))
g = VoxelGraph((n, n, n), connectivity=26)
g._set_boundary(s["term"], s["image"], s["sigma"], False)
if regional:
    rg = synthetic.regional((n, n, n))
    g._set_regional(rg["prob"], rg["alpha"])
g._set_markers(s["fg"], s["bg"])
g.set_param("max_outer", 1)
for rep in range(2):
    g._build()
    try:
        g.maxflow()
    except Exception as e:  # not converged: expected
        print("maxflow:", str(e)[:80])
