"""Quick GPU probe: times build+solve at a few sizes and prints solver stats (development aid)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from medpy_amd import synthetic
from medpy_amd.graphcut.graph import VoxelGraph

sizes = [int(a) for a in sys.argv[1:]] or [64, 128, 256]
for n in sizes:
    s = synthetic.sphere((n, n, n))
    g = VoxelGraph((n, n, n))
    g._set_boundary("difference_exponential", s["image"], s["sigma"], False)
    g._set_markers(s["fg"], s["bg"])
    for rep in range(2):
        t0 = time.perf_counter(); g._build(); t1 = time.perf_counter(); f = g.maxflow(); t2 = time.perf_counter()
        st = g.stats()
        print(json.dumps({"n": n, "rep": rep, "wall_build_ms": (t1 - t0) * 1e3, "wall_solve_ms": (t2 - t1) * 1e3,
                          "mvox_s": n ** 3 / (t2 - t0) / 1e6, "flow": f, "fg": float(g.labels().mean()), **st}), flush=True)
    g.set_param("kernel_timing", 0)
    t0 = time.perf_counter(); g._build(); f = g.maxflow(); t2 = time.perf_counter()
    print(json.dumps({"n": n, "untimed_kernels_total_ms": (t2 - t0) * 1e3, "mvox_s": n ** 3 / (t2 - t0) / 1e6}), flush=True)
    g.close()
