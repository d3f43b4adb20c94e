"""Schedule sweep for the 26-neighbourhood solver (development aid)"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd import synthetic
from medpy_amd.graphcut.graph import VoxelGraph
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = synthetic.sphere((n, n, n)); r = synthetic.regional((n, n, n))
g = VoxelGraph((n, n, n), connectivity=26)
g._set_boundary("difference_exponential", s["image"], s["sigma"], False)
g._set_markers(s["fg"], s["bg"])
g._set_regional(r["prob"], r["alpha"])
ref = None
for c, w, rr in ((1, 4, 8), (1, 3, 8), (1, 5, 8), (1, 2, 8), (1, 4, 6), (1, 4, 10), (1, 3, 6), (1, 3, 10)):
    g.set_param("max_cycles", c); g.set_param("max_sweeps", w); g.set_param("rounds_per_relabel", rr)
    best = 1e9
    for rep in range(2):
        t0 = time.perf_counter(); g._build(); f = g.maxflow(); best = min(best, time.perf_counter() - t0)
    lab = g.labels()
    if ref is None:
        ref = lab.copy()
    st = g.stats()
    print(json.dumps({"n": n, "cycles": c, "sweeps": w, "rounds": rr, "ms": round(best * 1e3, 2), "same": bool((lab == ref).all()),
                      "relabels": st["global_relabels"], "phases": st["phases"], "dis_tiles": st["discharge_tiles"], "rel_tiles": st["relabel_tiles"]}), flush=True)
