"""26-neighbourhood: incremental global relabel on / off (development aid).  python tools/gpu_26_incremental.py [n] [workload]"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd import synthetic
from medpy_amd.graphcut.graph import VoxelGraph
from medpy_amd._lib import VIOLATION_KEYS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
gen = sys.argv[2] if len(sys.argv) > 2 else "sphere"
for regional in (True, False):
    s = getattr(synthetic, gen)((n, n, n))
    g = VoxelGraph((n, n, n), connectivity=26)
    g._set_boundary(s["term"], s["image"], s["sigma"], False)
    g._set_markers(s["fg"], s["bg"])
    if regional:
        r = synthetic.regional((n, n, n))
        g._set_regional(r["prob"], r["alpha"])
    ref = None
    for inc in (0, 1):
        g.set_param("incremental_relabel", inc)
        best = 1e9
        for rep in range(2):
            t0 = time.perf_counter(); g._build(); f = g.maxflow(); dt = time.perf_counter() - t0
            if dt < best:
                best, st = dt, g.stats()
        lab = g.labels()
        ref = lab if ref is None else ref
        v = g.validate()
        print(json.dumps({"n": n, "wl": gen, "regional": regional, "incremental": inc, "ms": round(best * 1e3, 2), "same_labels": bool((lab == ref).all()), "flow": f,
                          "valid": not any(v[k] for k in VIOLATION_KEYS),
                          **{k: (round(st[k], 2) if isinstance(st[k], float) else st[k]) for k in ("build_ms", "discharge_ms", "relabel_ms", "global_relabels", "phases", "discharge_tiles", "relabel_tiles")}}), flush=True)
