"""How much of the relabel work is the FIRST (from-scratch) global relabel? (development aid)"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd import _lib, synthetic
from medpy_amd.graphcut.graph import VoxelGraph
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for gen in sys.argv[2:] or ["sphere"]:
    s = getattr(synthetic, gen)((n, n, n))
    g = VoxelGraph((n, n, n))
    g._set_boundary(s["term"], s["image"], s["sigma"], False)
    g._set_markers(s["fg"], s["bg"])
    for outer in (1, 2, 100000):
        g.set_param("max_outer", outer)
        g._build()
        try:
            g.maxflow()
        except _lib.MedpyHipError:
            pass
        st = g.stats()
        print(json.dumps({"gen": gen, "n": n, "max_outer": outer, **{k: st[k] for k in ("solve_ms", "discharge_ms", "relabel_ms", "relabel_launches", "relabel_tiles", "discharge_tiles", "global_relabels", "phases")}}), flush=True)
