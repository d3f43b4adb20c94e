"""26-neighbourhood (+ regional) timing probe (development aid)"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd import synthetic
from medpy_amd.graphcut.graph import VoxelGraph
for n in [int(a) for a in sys.argv[1:]] or [128, 256]:
    s = synthetic.sphere((n, n, n)); r = synthetic.regional((n, n, n))
    g = VoxelGraph((n, n, n), connectivity=26)
    g._set_boundary("difference_exponential", s["image"], s["sigma"], False)
    g._set_markers(s["fg"], s["bg"])
    g._set_regional(r["prob"], r["alpha"])
    for rep in range(2):
        t0 = time.perf_counter(); g._build(); t1 = time.perf_counter(); f = g.maxflow(); t2 = time.perf_counter()
        print(json.dumps({"n": n, "conn": 26, "regional": True, "build_ms": (t1 - t0) * 1e3, "solve_ms": (t2 - t1) * 1e3,
                          "mvox_s": n ** 3 / (t2 - t0) / 1e6, "flow": f, "fg": float(g.labels().mean()), **g.stats()}), flush=True)
    g.close()
