"""Where do k_discharge_w's cycles go? (development aid; needs a library built with -DMGCW_PROFILE:
bash tools/ab_variant.sh build prof -DMGCW_PROFILE ; MEDPY_HIP_LIB=build/lib_prof.so python tools/gpu_sections.py 512)"""
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
g.set_param("profile_sections", 1)
g._build(); t0 = time.perf_counter(); g.maxflow(); dt = time.perf_counter() - t0
st = g.stats(); pr = g.profile()
names = {"labels": "loop top: ticket, list entry, status word   (mark 1)",
         "votes": "loads issued; halo + inbox back, staged     (mark 4)",
         "faceflags": "own state back, inbox absorbed              (mark 5)",
         "load": "dirty votes, labels to LDS, register set-up (mark 0)",
         "sweep": "one sweep                                   (mark 2)",
         "s6": "face votes, claims out, tail votes, staging (mark 6)",
         "s7": "claims back, positions, write-back issued   (mark 7)",
         "store": "positions back, outbox + list entries out   (mark 3)"}
tot = sum(pr[k]["cycles"] for k in names)
print(json.dumps({"n": n, "solve_ms": dt * 1e3, "discharge_tiles": st["discharge_tiles"]}))
for k, label in names.items():
    v = pr[k]
    print("%-58s cycles %16d (%5.1f%%)  count %9d  avg %8.0f cycles" % (label, v["cycles"], 100.0 * v["cycles"] / max(tot, 1), v["count"], v["cycles"] / max(v["count"], 1)))
print("per tile discharge: %.0f cycles; sweeps per discharge %.2f" % (tot / max(st["discharge_tiles"], 1), pr["sweep"]["count"] / max(st["discharge_tiles"], 1)))
