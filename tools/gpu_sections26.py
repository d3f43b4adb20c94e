"""Where do k26_discharge's cycles go? (development aid)   python tools/gpu_sections26.py [n] [regional 0|1]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd import synthetic
from medpy_amd.graphcut.graph import VoxelGraph
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
regional = (sys.argv[2] if len(sys.argv) > 2 else "1") == "1"
s = synthetic.sphere((n, n, n))
g = VoxelGraph((n, n, n), connectivity=26)
g._set_boundary("difference_exponential", s["image"], s["sigma"], False)
g._set_markers(s["fg"], s["bg"])
if regional:
    r = synthetic.regional((n, n, n))
    g._set_regional(r["prob"], r["alpha"])
g._build(); g.maxflow()
t0 = time.perf_counter(); g._build(); g.maxflow(); plain = time.perf_counter() - t0
g.set_param("profile_sections", 1)
g._build(); t0 = time.perf_counter(); g.maxflow(); dt = time.perf_counter() - t0
st = g.stats(); pr = g.profile()
names = {"load": "load", "labels": "label set-up / between tiles", "faceflags": "sweep: direction mask", "s6": "sweep: the steps", "sweep": "sweep: local relabel", "store": "tail votes + stores"}
tot = sum(pr[k]["cycles"] for k in names)
print(json.dumps({"n": n, "regional": regional, "solve_ms": plain * 1e3, "profiled_ms": dt * 1e3, "discharge_tiles": st["discharge_tiles"], "discharge_ms": st["discharge_ms"],
                  "relabel_ms": st["relabel_ms"], "build_ms": st["build_ms"], "phases": st["phases"], "relabels": st["global_relabels"]}))
for k, label in names.items():
    v = pr[k]
    print("%-32s cycles %16d (%5.1f%%)  count %9d  avg %8.0f" % (label, v["cycles"], 100.0 * v["cycles"] / max(tot, 1), v["count"], v["cycles"] / max(v["count"], 1)))
print("per tile discharge: %.0f clock64 ticks; sweeps per discharge %.2f" % (tot / max(st["discharge_tiles"], 1), pr["sweep"]["count"] / max(st["discharge_tiles"], 1)))
if len(sys.argv) > 3:  # one more run with extra parameters, e.g. sweeps_sparse26=8
    pass
