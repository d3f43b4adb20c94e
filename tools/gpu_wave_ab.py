"""A/B of the kernel forms and schedule knobs on the GPU (development aid).

    python tools/gpu_wave_ab.py [n] [workload] [variant ...]     variant = wave:sweeps:rounds:grid:adaptive  (0 = default; adaptive 1 = off, k + 1 = threshold k (default 3))

Every variant must return the labels of the first one; prints one JSON line per variant."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd import synthetic
from medpy_amd.graphcut.graph import VoxelGraph

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
gen = sys.argv[2] if len(sys.argv) > 2 else "sphere"
variants = sys.argv[3:] or ["0:0:0", "1:0:0", "3:0:0", "7:0:0"]
s = getattr(synthetic, gen)((n, n, n))
g = VoxelGraph((n, n, n))
g._set_boundary("difference_exponential", s["image"], s["sigma"], False)
g._set_markers(s["fg"], s["bg"])
ref = None
for v in variants:
    f = [int(x) for x in v.split(":")] + [0, 0, 0, 0, 0]
    g.set_param("wave_kernels", f[0])
    g.set_param("max_sweeps", f[1] or 12)
    g.set_param("rounds_per_relabel", f[2] or 8)
    if f[3]:
        g.set_param("wave_grid_dis", f[3])
    if f[4]:
        g.set_param("adaptive_rounds", f[4] - 1)
    best, bst = 1e9, None
    for rep in range(3):
        t0 = time.perf_counter(); g._build(); fl = g.maxflow(); dt = time.perf_counter() - t0
        if dt < best:
            best, bst = dt, g.stats()
    lab = g.labels()
    if ref is None:
        ref = lab.copy()
    st = bst
    print(json.dumps({"n": n, "wl": gen, "variant": v, "ms": round(best * 1e3, 2), "mvox_s": round(n ** 3 / best / 1e6, 1),
                      "same_labels": bool((lab == ref).all()), "flow": fl, "build_ms": round(st["build_ms"], 2),
                      "discharge_ms": round(st["discharge_ms"], 2), "relabel_ms": round(st["relabel_ms"], 2),
                      "relabels": st["global_relabels"], "phases": st["phases"], "dis_tiles": st["discharge_tiles"],
                      "rel_tiles": st["relabel_tiles"], "rel_launches": st["relabel_launches"], "readbacks": st["readbacks"]}), flush=True)
