"""use_filters A/B (development aid)"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd import synthetic
from medpy_amd.graphcut.graph import VoxelGraph
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = synthetic.sphere((n, n, n))
g = VoxelGraph((n, n, n))
g._set_boundary("difference_exponential", s["image"], s["sigma"], False)
g._set_markers(s["fg"], s["bg"])
ref = None
for uf in (3, 7, 3, 7):
    g.set_param("use_filters", uf)
    t0 = time.perf_counter(); g._build(); f = g.maxflow(); dt = time.perf_counter() - t0
    st = g.stats(); lab = g.labels()
    if ref is None: ref = lab.copy()
    print(json.dumps({"use_filters": uf, "ms": round(dt * 1e3, 2), "same": bool((lab == ref).all()), "flow": f, **{k: st[k] for k in ("global_relabels", "phases", "discharge_tiles", "relabel_tiles", "relabel_launches")}}), flush=True)
