"""Default-schedule timing on the three synthetic workloads (development aid): sphere (headline), hard (weak contrast),
ties (integer image, tie-degenerate)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd import synthetic
from medpy_amd.graphcut.graph import VoxelGraph
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for gen in ("sphere", "hard", "ties"):
    s = getattr(synthetic, gen)((n, n, n))
    g = VoxelGraph((n, n, n))
    g._set_boundary(s["term"], s["image"], s["sigma"], False)
    g._set_markers(s["fg"], s["bg"])
    best = 1e9
    for rep in range(2):
        t0 = time.perf_counter(); g._build(); f = g.maxflow(); best = min(best, time.perf_counter() - t0)
    st = g.stats()
    print(json.dumps({"gen": gen, "n": n, "ms": round(best * 1e3, 2), "mvox_s": round(n ** 3 / best / 1e6, 1), "flow": f, "fg": float(g.labels().mean()),
                      **{k: st[k] for k in ("global_relabels", "phases", "discharge_tiles", "relabel_tiles")}}), flush=True)
    g.close()
