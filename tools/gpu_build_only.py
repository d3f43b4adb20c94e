"""k_build alone (development aid): device time of mgc_build at n^3 for the libraries named on the command line (MEDPY_HIP_LIB per process)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from medpy_amd import synthetic
    from medpy_amd.graphcut.graph import VoxelGraph
    n = int(sys.argv[2])
    s = synthetic.sphere((n, n, n))
    g = VoxelGraph((n, n, n))
    g._set_boundary(sys.argv[3], s["image"], s["sigma"], False)
    g._set_markers(s["fg"], s["bg"])
    ms = []
    for _ in range(5):
        g._build()
        ms.append(g.stats()["build_ms"])
    print(json.dumps({"lib": os.environ.get("MEDPY_HIP_LIB", "tree"), "n": n, "term": sys.argv[3], "build_ms": [round(v, 3) for v in ms]}))
else:
    n = sys.argv[1]
    for lib in sys.argv[2:] or ["tree"]:
        env = dict(os.environ)
        if lib != "tree":
            env["MEDPY_HIP_LIB"] = os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", n, "difference_exponential"], env=env, capture_output=True, text=True, timeout=300)
        print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1])
