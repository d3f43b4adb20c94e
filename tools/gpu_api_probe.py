"""Where the time of one volume through the public API goes (development aid): handle, H2D, build, solve, read-out, close."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from medpy_amd import graphcut, synthetic  # noqa: E402
from medpy_amd.graphcut.graph import VoxelGraph  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
s = synthetic.sphere((n, n, n))
for rep in range(3):
    t = [time.perf_counter()]
    g = VoxelGraph((n, n, n)); t.append(time.perf_counter())
    g._set_boundary(s["term"], s["image"], s["sigma"], False); t.append(time.perf_counter())
    g._set_markers(s["fg"], s["bg"]); t.append(time.perf_counter())
    g._build(); t.append(time.perf_counter())
    g.maxflow(); t.append(time.perf_counter())
    lab = g.labels(); t.append(time.perf_counter())
    g.close(); t.append(time.perf_counter())
    names = ("handle", "set_boundary", "set_markers", "build", "maxflow", "labels", "close")
    print(json.dumps({"rep": rep, "total_ms": round((t[-1] - t[0]) * 1e3, 2), **{k: round((b - a) * 1e3, 2) for k, a, b in zip(names, t, t[1:])}}))
for rep in range(3):
    t0 = time.perf_counter()
    g = graphcut.graph_from_voxels(s["fg"], s["bg"], boundary_term=graphcut.energy_voxel.boundary_difference_exponential, boundary_term_args=(s["image"], s["sigma"], False))
    t1 = time.perf_counter(); g.maxflow(); t2 = time.perf_counter(); g.labels(); t3 = time.perf_counter(); g.close(); t4 = time.perf_counter()
    print(json.dumps({"api_rep": rep, "total_ms": round((t4 - t0) * 1e3, 2), "graph_from_voxels": round((t1 - t0) * 1e3, 2), "maxflow": round((t2 - t1) * 1e3, 2), "labels": round((t3 - t2) * 1e3, 2), "close": round((t4 - t3) * 1e3, 2)}))
