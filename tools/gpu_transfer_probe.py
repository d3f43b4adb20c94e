"""Host <-> HBM transfer rates of the C-ABI boundary (mgc_set_boundary / mgc_set_markers / mgc_labels) for a few settings of
MEDPY_HIP_STAGE_THREADS (development aid; one process per setting: the setting is read once)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np
    from medpy_amd import synthetic
    from medpy_amd.graphcut.graph import VoxelGraph
    n = 512
    s = synthetic.sphere((n, n, n))
    g = VoxelGraph((n, n, n))
    up, down = [], []
    for rep in range(3):
        t0 = time.perf_counter()
        g._set_boundary(s["term"], s["image"], s["sigma"], False)
        g._set_markers(s["fg"], s["bg"])
        up.append(time.perf_counter() - t0)
        g._build(); g.maxflow()
        g._labels = None
        t0 = time.perf_counter()
        g.labels()
        down.append(time.perf_counter() - t0)
    print(json.dumps({"threads": os.environ.get("MEDPY_HIP_STAGE_THREADS"), "h2d_ms": round(min(up) * 1e3, 2), "h2d_gbs": round(0.805306368 / min(up), 1),
                      "d2h_ms": round(min(down) * 1e3, 2), "d2h_gbs": round(0.134217728 / min(down), 1)}))
else:
    for t in (sys.argv[1:] or ["0", "2", "4", "8"]):
        env = dict(os.environ, MEDPY_HIP_STAGE_THREADS=t)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, capture_output=True, text=True, timeout=300)
        print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1])
