"""Schedule-knob sweep on the GPU (development aid): one-factor-at-a-time around the defaults."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd import synthetic
from medpy_amd.graphcut.graph import VoxelGraph

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
gen = sys.argv[2] if len(sys.argv) > 2 else "sphere"
s = getattr(synthetic, gen)((n, n, n))
g = VoxelGraph((n, n, n))
g._set_boundary("difference_exponential", s["image"], s["sigma"], False)
g._set_markers(s["fg"], s["bg"])
base = dict(rounds_per_relabel=12, max_cycles=4, max_sweeps=8, relabel_batch=8, check_rounds=4, grid_cap=4096)
grid = [dict()]
for k, vals in dict(rounds_per_relabel=[2, 3, 4, 6, 8, 16], max_cycles=[1, 2, 3, 6, 8], max_sweeps=[2, 4, 6, 12, 16],
                    check_rounds=[1, 2, 8], grid_cap=[1024, 2048, 8192]).items():
    grid += [{k: v} for v in vals]
grid += [dict(rounds_per_relabel=4, max_cycles=2), dict(rounds_per_relabel=4, max_cycles=2, max_sweeps=4), dict(rounds_per_relabel=6, max_cycles=2, max_sweeps=6),
         dict(rounds_per_relabel=3, max_cycles=2, max_sweeps=4, check_rounds=1), dict(rounds_per_relabel=6, max_cycles=3, max_sweeps=4),
         dict(rounds_per_relabel=8, max_cycles=2, max_sweeps=4), dict(rounds_per_relabel=4, max_cycles=3, max_sweeps=6, check_rounds=2)]
g.set_param("kernel_timing", 0)
ref = None
for over in grid:
    cfg = dict(base); cfg.update(over)
    for k, v in cfg.items():
        g.set_param(k, v)
    best = 1e9
    for rep in range(2):
        t0 = time.perf_counter(); g._build(); f = g.maxflow(); dt = time.perf_counter() - t0
        best = min(best, dt)
    st = g.stats()
    lab = g.labels()
    if ref is None:
        ref = lab.copy()
    same = bool((lab == ref).all())
    print(json.dumps({"n": n, "over": over, "ms": round(best * 1e3, 2), "mvox_s": round(n ** 3 / best / 1e6, 1), "same_labels": same,
                      "relabels": st["global_relabels"], "phases": st["phases"], "dis_tiles": st["discharge_tiles"],
                      "rel_tiles": st["relabel_tiles"], "readbacks": st["readbacks"]}), flush=True)
