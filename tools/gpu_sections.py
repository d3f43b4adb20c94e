"""Where do k_discharge's cycles go? (development aid)"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd import synthetic
from medpy_amd.graphcut.graph import VoxelGraph
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = synthetic.sphere((n, n, n))
g = VoxelGraph((n, n, n))
g._set_boundary("difference_exponential", s["image"], s["sigma"], False)
g._set_markers(s["fg"], s["bg"])
g._build(); g.maxflow()
if len(sys.argv) > 2:
    g.set_param("grid_cap", int(sys.argv[2]))
g.set_param("profile_sections", 1)
g._build(); t0 = time.perf_counter(); g.maxflow(); dt = time.perf_counter() - t0
st = g.stats(); pr = g.profile()
tot = sum(v["cycles"] for v in pr.values())
print(json.dumps({"n": n, "solve_ms": dt * 1e3, "stats": st}))
for k, v in pr.items():
    print("%-7s cycles %14d (%.1f%%)  count %9d  avg %.0f cycles" % (k, v["cycles"], 100.0 * v["cycles"] / max(tot, 1), v["count"], v["cycles"] / max(v["count"], 1)))
print("per tile discharge: %.0f cycles; labels per discharge %.2f; sweeps per discharge %.2f" % (
    tot / max(st["discharge_tiles"], 1), pr["labels"]["count"] / max(st["discharge_tiles"], 1), pr["sweep"]["count"] / max(st["discharge_tiles"], 1)))
