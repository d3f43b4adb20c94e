#!/usr/bin/env python3
"""Work profile of the lattice solver from the host simulator (CPU only, development aid).

    python tools/sim_workprofile.py [n] [workload]

Counts tile discharges / sweeps / relabels and where the discharges happen (inside / outside the bright ball),
so schedule changes can be compared without a GPU (same tile operations as the HIP kernels).
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
import sim  # noqa: E402
from medpy_amd import synthetic  # noqa: E402
from oracle import energy_numpy  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    wl = sys.argv[2] if len(sys.argv) > 2 else "sphere"
    mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    s = getattr(synthetic, wl)((n, n, n))
    w = energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"])
    tr = np.where(s["fg"], 65535.0, 0.0) - np.where(s["bg"], 65535.0, 0.0)
    L = sim.lib()
    g = (n + 7) // 8
    nt = g ** 3
    prof = np.zeros(64, np.int64)
    tiles = np.zeros(nt, np.int32)
    L.hostsim_prof.argtypes = [np.ctypeslib.ndpointer(np.int64), np.ctypeslib.ndpointer(np.int32), C.c_int]
    L.hostsim_prof(prof, tiles, nt)  # arm
    L.hostsim_set_wave_mode(mode)
    t0 = time.time()
    labels, st = sim.solve((n, n, n), w, tr, sweeps=sweeps, rounds=rounds)
    L.hostsim_set_wave_mode(0)
    if os.environ.get("CHECK"):
        from oracle import pipeline
        ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term=s["term"], image=s["image"], sigma=s["sigma"])
        print("labels differ from BK in", int((labels.astype(bool) != ref.labels.astype(bool)).sum()))
    dt = time.time() - t0
    L.hostsim_prof(prof, tiles, nt)
    tiles = tiles.reshape(g, g, g)
    c = (np.arange(g) * 8 + 3.5) - (n - 1) / 2.0
    r = np.sqrt(c[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2)
    inside = r < 0.3 * n - 7
    shell = (r >= 0.3 * n - 7) & (r <= 0.3 * n + 7)
    outside = r > 0.3 * n + 7
    print("n=%d %s  sim %.1fs  fg=%.5f" % (n, wl, dt, labels.mean()))
    print("stats", st)
    print("wave steps: lanes %d any %d shift %d (per discharge %.0f / %.0f / %.0f)" % (
        prof[20], prof[21], prof[22], prof[20] / max(1, prof[3]), prof[21] / max(1, prof[3]), prof[22] / max(1, prof[3])))
    print("wave forms through w.ld / w.st: %.1f KB read, %.1f KB written per discharge" % (prof[50] / 1024.0 / max(1, prof[3]), prof[51] / 1024.0 / max(1, prof[3])))
    print("discharges %d  label sections %d  sweeps %d (%.2f/discharge)  waves active %.3f" % (
        prof[3], prof[1], prof[2], prof[2] / max(1, prof[3]), prof[16] / max(1, prof[17])))
    for name, m in (("inside", inside), ("shell", shell), ("outside", outside)):
        print("  %-8s tiles %6d  discharges %8d  (%.2f per tile)" % (name, m.sum(), tiles[m].sum(), tiles[m].sum() / max(1, m.sum())))
    print("  per-voxel: discharges %.2f relabels %.2f" % (st["discharge_tiles"] / nt, st["relabel_tiles"] / nt))
    if prof[24]:
        print("incremental relabels reset %d tiles: %.1f%% got their labels back unchanged, %.1f%% unchanged on all six faces; "
              "%.1f voxels changed per reset tile" % (prof[24], 100.0 * prof[25] / prof[24], 100.0 * prof[26] / prof[24], prof[27] / prof[24]))


if __name__ == "__main__":
    main()
