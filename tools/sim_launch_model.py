#!/usr/bin/env python3
"""How much of a discharge launch is the ragged end?  (CPU only, development aid.)

    python tools/sim_launch_model.py [n] [workload] [tiles_per_wave]

Runs the wave-form solver in the host simulator with its discharge trace on (one record per tile visit: phase, tile,
sweeps), then schedules every phase's visits greedily onto P = visits / tiles_per_wave persistent waves with the cost
model of the section profile (43.6 k cycles per visit + 7.0 k per sweep, profiles/README.md) and compares

    ideal       sum of the costs / P (perfectly divisible work)
    random      the order the list happens to have (what the ticket counter of k_discharge_w sees)
    lpt_oracle  longest visit first, with the sweeps known in advance
    lpt_last    longest first, predicted by the sweeps of the tile's previous visit

At 2.6 visits per wave (512^3: ~5.4 k tiles per launch on 2048 resident waves) the quantisation of the visit costs,
not the order, is what separates `random` from `ideal`: sorting by predicted cost was measured here before it was written
for the GPU, and not written.
"""
import collections
import ctypes as C
import heapq
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
import sim  # noqa: E402
from medpy_amd import synthetic  # noqa: E402
from oracle import energy_numpy  # noqa: E402

FIXED, PER_SWEEP = 43.6, 7.0  # k cycles


def trace(n, wl, path):
    s = getattr(synthetic, wl)((n, n, n))
    w = energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"])
    tr = np.where(s["fg"], 65535.0, 0.0) - np.where(s["bg"], 65535.0, 0.0)
    L = sim.lib()
    L.hostsim_trace.argtypes = [C.c_char_p]
    L.hostsim_trace(path.encode())
    try:
        _, st = sim.solve((n, n, n), w, tr, wave_mode=1)
    finally:
        L.hostsim_trace(b"")
    return st


def makespan(costs, slots):
    h = [0.0] * slots
    heapq.heapify(h)
    for c in costs:
        heapq.heappush(h, heapq.heappop(h) + c)
    return max(h)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    wl = sys.argv[2] if len(sys.argv) > 2 else "sphere"
    ratio = float(sys.argv[3]) if len(sys.argv) > 3 else 2.63
    path = "/tmp/hostsim_trace_%s_%d.txt" % (wl, n)
    print("solve:", trace(n, wl, path))
    phases = collections.OrderedDict()
    for line in open(path):
        p, t, s = (int(v) for v in line.split())
        phases.setdefault(p, []).append((t, s))
    total, hist, last = collections.Counter(), collections.Counter(), {}
    for p, visits in phases.items():
        slots = max(1, int(round(len(visits) / ratio)))
        costs = [FIXED + PER_SWEEP * s for _, s in visits]
        shuffled = costs[:]
        random.Random(p).shuffle(shuffled)
        total["ideal"] += sum(costs) / slots
        total["random"] += makespan(shuffled, slots)
        total["lpt_oracle"] += makespan(sorted(costs, reverse=True), slots)
        guess = sorted(visits, key=lambda ts: -last.get(ts[0], 6))
        total["lpt_last"] += makespan([FIXED + PER_SWEEP * s for _, s in guess], slots)
        for t, s in visits:
            hist[s] += 1
            last[t] = s
    print("k cycles over all launches:", {k: round(v, 1) for k, v in total.items()})
    print("visits by sweeps:", sorted(hist.items()))


if __name__ == "__main__":
    main()
