"""Where do k26_discharge_w's cycles go? (development aid; needs a library built with -DMGCW_PROFILE:
bash tools/ab_variant.sh build prof -DMGCW_PROFILE ; MEDPY_HIP_LIB=build/lib_prof.so python tools/gpu_sections26w.py 512 [regional] [name=value,...])"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd import synthetic
from medpy_amd.graphcut.graph import VoxelGraph
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
regional = len(sys.argv) > 2 and sys.argv[2] == "regional"
params = sys.argv[3] if len(sys.argv) > 3 else "wave_kernels=41"
s = synthetic.sphere((n, n, n))
g = VoxelGraph((n, n, n), connectivity=26)
g._set_boundary("difference_exponential", s["image"], s["sigma"], False)
if regional:
    r = synthetic.regional((n, n, n))
    g._set_regional(r["prob"], r["alpha"])
g._set_markers(s["fg"], s["bg"])
for kv in params.split(","):
    k, v = kv.split("=")
    g.set_param(k, int(v))
g._build(); g.maxflow()
g.set_param("profile_sections", 1)
g._build(); t0 = time.perf_counter(); g.maxflow(); dt = time.perf_counter() - t0
st = g.stats(); pr = g.profile()
names = {"s7": "between two tiles: ticket, list entry        (mark 7)",
         "load": "halo + state loaded, LDS staged              (mark 0)",
         "labels": "pass A: push masks from the labels           (mark 1)",
         "faceflags": "one (slot, direction) step that ran          (mark 5)",
         "s6": "a slot's flush + skipped steps behind it     (mark 6)",
         "sweep": "rest of the passes over the steps            (mark 2)",
         "store": "pass R: local relabel                        (mark 3)",
         "votes": "tail: wake-ups, write-back                   (mark 4)"}
tot = sum(pr[k]["cycles"] for k in names)
print(json.dumps({"n": n, "regional": regional, "params": params, "solve_ms": dt * 1e3, "discharge_tiles": st["discharge_tiles"], "discharge_ms": st["discharge_ms"], "phases": st["phases"]}))
for k, label in names.items():
    v = pr[k]
    print("%-58s cycles %16d (%5.1f%%)  count %9d  avg %8.0f cycles" % (label, v["cycles"], 100.0 * v["cycles"] / max(tot, 1), v["count"], v["cycles"] / max(v["count"], 1)))
print("per tile discharge: %.0f cycles; sweeps per discharge %.2f" % (tot / max(st["discharge_tiles"], 1), pr["labels"]["count"] / max(st["discharge_tiles"], 1)))
