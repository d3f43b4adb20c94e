#!/usr/bin/env python3
"""Measurement of the region graph cut on one MI355X: graph_from_labels + boundary_stawiaski + maxflow
for a synthetic volume cut into cubic super-voxels, next to the CPU oracle (NumPy restatement of the term + the reference
BK core).  Prints one JSON line.  Usage: python tools/bench_labels.py [--n 256] [--block 4] [--no-cpu]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--block", type=int, default=4)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    from medpy_amd import graphcut, synthetic
    from medpy_amd.graphcut import energy_label as el
    n, shape = a.n, (a.n,) * 3
    s = synthetic.sphere(shape)
    idx = np.indices(shape)
    coarse = tuple((idx[d] + (idx[(d + 1) % 3] // 9)) // a.block for d in range(3))  # ragged super-voxels
    flat = np.ravel_multi_index(coarse, [int(c.max()) + 1 for c in coarse])
    _, lab = np.unique(flat, return_inverse=True)
    lab = (lab.reshape(shape) + 1).astype(np.int32)
    del idx, coarse, flat
    grad = np.abs(np.gradient(s["image"].astype(np.float32))[0]).astype(np.float32)
    nreg = int(lab.max())
    times = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        g = graphcut.graph_from_labels(lab, s["fg"], s["bg"], boundary_term=el.boundary_stawiaski, boundary_term_args=grad)
        t1 = time.perf_counter()
        flow = g.maxflow()
        t2 = time.perf_counter()
        seg = g.labels()
        st = g.stats()
        times.append((t1 - t0, t2 - t1))
        del g
    build_s, solve_s = min(t[0] for t in times), min(t[1] for t in times)
    out = {"workload": "%d^3 volume, %d super-voxel regions (block %d, ragged), boundary_stawiaski" % (n, nreg, a.block),
           "regions": nreg, "arcs": st["arcs"], "border_pixel_pairs": st["edges_added"],
           "gpu_graph_from_labels_s": round(build_s, 4), "gpu_maxflow_s": round(solve_s, 4),
           "gpu_maxflow_device_ms": {"csr_build": round(st["build_ms"], 3), "solve": round(st["solve_ms"], 3)},
           "rounds": st["rounds"], "global_relabels": st["global_relabels"], "flow": flow, "fg_regions": int(seg.sum()),
           "mvoxels_per_s": round(n ** 3 / (build_s + solve_s) / 1e6, 2)}
    if not a.no_cpu:
        from oracle import energy_label_numpy as eln
        t0 = time.perf_counter()
        og = eln.build_label_graph(lab, s["fg"], s["bg"], "stawiaski", grad)
        t1 = time.perf_counter()
        oflow = og.maxflow()
        t2 = time.perf_counter()
        out["cpu_oracle"] = {"build_s": round(t1 - t0, 3), "maxflow_s": round(t2 - t1, 4), "kind": "NumPy restatement + reference BK core, 1 core",
                             "flow": oflow, "labels_equal": bool((og.labels().astype(bool) == seg).all())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
