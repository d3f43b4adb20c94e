"""A/B of schedule / kernel-form parameters on the GPU (development aid).

    python tools/gpu_ab.py [--n 512] [--wl sphere] [--conn 6] [--reps 3] [--lib path.so] variant [variant ...]

variant = comma separated name=value pairs for mgc_set_param ("base" = defaults), e.g.  base  first_relabel_dt=0  max_sweeps=8,rounds_per_relabel=10
Every variant must return the labels of the first one.  One JSON line per variant (best of --reps build + solve)."""
import argparse
import json
import os
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--wl", default="sphere")
ap.add_argument("--conn", type=int, default=6)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--lib", default=None)
ap.add_argument("--tag", default="")
ap.add_argument("--regional", action="store_true", help="add the regional_probability_map term of BASELINE config 3")
ap.add_argument("--term", default=None, help="boundary term instead of the workload's own (e.g. difference_division: what k_build costs without exp)")
ap.add_argument("variants", nargs="*")
a = ap.parse_args()
if a.lib:
    os.environ["MEDPY_HIP_LIB"] = a.lib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from medpy_amd import synthetic  # noqa: E402
from medpy_amd.graphcut.graph import VoxelGraph  # noqa: E402

n = a.n
s = getattr(synthetic, a.wl)((n, n, n))
ref = None
for v in (a.variants or ["base"]):
    g = VoxelGraph((n, n, n), connectivity=a.conn)  # a fresh handle per variant: defaults restored
    g._set_boundary(a.term or s["term"], s["image"], s["sigma"], False)
    if a.regional:
        rg = synthetic.regional((n, n, n))
        g._set_regional(rg["prob"], rg["alpha"])
    g._set_markers(s["fg"], s["bg"])
    if v != "base":
        for kv in v.split(","):
            k, val = kv.split("=")
            g.set_param(k, int(val))
    best, bst, fl = 1e9, None, None
    for rep in range(a.reps):
        t0 = time.perf_counter()
        g._build()
        fl = g.maxflow()
        dt = time.perf_counter() - t0
        if dt < best:
            best, bst = dt, g.stats()
    lab = g.labels()
    if ref is None:
        ref = lab.copy()
    st = bst
    print(json.dumps({"tag": a.tag, "n": n, "wl": a.wl, "conn": a.conn, "regional": a.regional, "variant": v, "ms": round(best * 1e3, 2), "mvox_s": round(n ** 3 / best / 1e6, 1),
                      "same_labels": bool((lab == ref).all()), "flow": fl, "build_ms": round(st["build_ms"], 2), "solve_ms": round(st["solve_ms"], 2),
                      "discharge_ms": round(st["discharge_ms"], 2), "relabel_ms": round(st["relabel_ms"], 2),
                      "relabels": st["global_relabels"], "phases": st["phases"], "dis_tiles": st["discharge_tiles"],
                      "rel_tiles": st["relabel_tiles"], "dis_launches": st["discharge_launches"], "rel_launches": st["relabel_launches"],
                      "readbacks": st["readbacks"]}), flush=True)
    del g
