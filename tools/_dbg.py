import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from medpy_amd import synthetic
from medpy_amd.graphcut.graph import VoxelGraph
n = int(sys.argv[1])
s = synthetic.sphere((n, n, n))
for pair in (0, 1):
    g = VoxelGraph((n, n, n))
    g._set_boundary(s["term"], s["image"], s["sigma"], False)
    g._set_markers(s["fg"], s["bg"])
    g.set_param("max_outer", 12)
    g.set_param("wave_min_tiles", 0)
    if pair:
        g.set_param("pair_phases", 1); g.set_param("pair_min_tiles", 0)
    g._build()
    try:
        f = g.maxflow(); print("pair", pair, "flow", f)
    except Exception as e:
        print("pair", pair, "failed:", str(e)[:80])
    st = g.stats()
    print({k: st[k] for k in ("global_relabels", "phases", "discharge_tiles", "relabel_tiles", "discharge_launches", "discharge_wave_tiles", "readbacks")})
    del g
