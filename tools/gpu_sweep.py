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
base = dict(rounds_per_relabel=8, max_cycles=1, max_sweeps=12, relabel_batch=8, check_rounds=4, grid_cap=4096)
grid = [dict()]
grid += [dict(max_sweeps=w) for w in (8, 16)]
grid += [dict(rounds_per_relabel=r) for r in (6, 10, 12)]
grid += [dict(grid_cap=c) for c in (2048, 8192)]
grid += [dict(rounds_per_relabel=10, max_sweeps=8), dict(rounds_per_relabel=12, max_sweeps=16), dict(relabel_batch=16), dict(check_rounds=8)]
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
