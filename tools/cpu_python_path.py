"""BASELINE config 1: the reference's as-shipped Python path (graph_from_voxels with its per-edge set_nweight loop, BK
maxflow, the per-voxel what_segment loop of bin/medpy_graphcut_voxel.py:177-181) on a 64^3 synthetic float32 volume, run
through the reference's OWN .py files (oracle/overlay.py) in the container that holds /root/reference.  The result is
committed as profiles/cpu_python_path_64.json; bench.py reports it next to the bulk C++ baseline it measures live."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from medpy_amd import synthetic  # noqa: E402
from oracle.overlay import import_reference_graphcut  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
gc = import_reference_graphcut()
s = synthetic.sphere((n, n, n))
best = None
for _ in range(3):
    t0 = time.perf_counter()
    g = gc.graph_from_voxels(s["fg"], s["bg"], boundary_term=gc.energy_voxel.boundary_difference_exponential,
                             boundary_term_args=(s["image"], s["sigma"], False))
    t1 = time.perf_counter()
    flow = g.maxflow()
    t2 = time.perf_counter()
    res = np.zeros(s["bg"].size, dtype=np.bool_)
    for idx in range(len(res)):
        res[idx] = 0 if g.termtype.SINK == g.what_segment(idx) else 1
    t3 = time.perf_counter()
    if best is None or t3 - t0 < best[0]:
        best = (t3 - t0, t1 - t0, t2 - t1, t3 - t2)
out = {"value": round(n ** 3 / best[0] / 1e6, 5), "unit": "Mvoxels/s", "cores": 1, "kind": "reference (its own Python, per-edge loop)",
       "workload": "%d^3 sphere volume (float32), 6-conn, boundary_difference_exponential sigma 15 (BASELINE config 1)" % n,
       "seconds": {"graph_from_voxels": round(best[1], 3), "maxflow": round(best[2], 4), "what_segment_loop": round(best[3], 3)},
       "flow": flow, "host": "build container, %d cores" % os.cpu_count(), "measured_with": "tools/cpu_python_path.py"}
json.dump(out, open(os.path.join(ROOT, "profiles", "cpu_python_path_64.json"), "w"), indent=1)
print(json.dumps(out))
