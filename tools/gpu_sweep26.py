"""Schedule sweep of the 26-neighbourhood solver on BASELINE config 3 (development aid): cycles < 0 = stored labels."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd import synthetic
from medpy_amd.graphcut.graph import VoxelGraph
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
regional = os.environ.get("REGIONAL", "1") == "1"
s = synthetic.sphere((n, n, n)); r = synthetic.regional((n, n, n))
g = VoxelGraph((n, n, n), connectivity=26)
g._set_boundary("difference_exponential", s["image"], s["sigma"], False)
g._set_markers(s["fg"], s["bg"])
if regional:
    g._set_regional(r["prob"], r["alpha"])
ref = None
if os.environ.get("GRID26"):
    g.set_param("grid26_dis", int(os.environ["GRID26"]))
for c, w, rr in [tuple(int(v) for v in a.split(":")) for a in sys.argv[2:]] or [(1, 3, 6), (-1, 3, 6), (-1, 6, 6), (1, 6, 6)]:
    g.set_param("max_cycles", c); g.set_param("max_sweeps", w); g.set_param("rounds_per_relabel", rr)
    best = 1e9
    for rep in range(2):
        t0 = time.perf_counter(); g._build(); f = g.maxflow(); best = min(best, time.perf_counter() - t0)
    st = g.stats(); lab = g.labels()
    ref = lab if ref is None else ref
    print(json.dumps({"n": n, "cycles": c, "sweeps": w, "rounds": rr, "ms": round(best * 1e3, 2), "same_labels": bool((lab == ref).all()), "flow": f,
                      **{k: (round(st[k], 2) if isinstance(st[k], float) else st[k]) for k in ("build_ms", "discharge_ms", "relabel_ms", "global_relabels", "phases", "discharge_tiles", "relabel_tiles")}}), flush=True)
