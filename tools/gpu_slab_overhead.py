"""the multi-GPU workload family on ONE GPU: how hard is the volume itself, how much does the slab driver add? (development aid)

    python tools/gpu_slab_overhead.py Z XY            every configuration in a process of its own (a handle of this size that
                                                      was created and destroyed before leaves the next one measurably slower)
"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from medpy_amd.graphcut.graph import VoxelGraph  # noqa: E402

Z, XY = int(sys.argv[1]), int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "all"
if mode == "all":
    for m in ("handle", "1", "2"):
        subprocess.run([sys.executable, os.path.abspath(__file__), str(Z), str(XY), m], check=False)
    sys.exit(0)
img, fg, bg = bench.block_volume(0, Z, Z // 512, XY // 512, 512)
if mode == "handle":
    g = VoxelGraph(img.shape)
    g._set_boundary("difference_exponential", img, 15.0, False)
    g._set_markers(fg, bg)
    for rep in range(2):
        t0 = time.perf_counter(); g._build(); f = g.maxflow(); dt = time.perf_counter() - t0
        st = g.stats()
        print(json.dumps({"shape": list(img.shape), "path": "single handle, mgc_maxflow", "ms": round(dt * 1e3, 1), "mvox_s": round(img.size / dt / 1e6, 1), "flow": f,
                          **{k: (round(st[k], 1) if isinstance(st[k], float) else st[k]) for k in ("build_ms", "discharge_ms", "relabel_ms", "global_relabels", "phases", "discharge_tiles", "relabel_tiles", "readbacks")}}), flush=True)
    g.close()
else:
    from medpy_amd.slab import HipSlab, LoopbackExchange, solve_slabs
    nslabs = int(mode)
    slabs = [HipSlab(img.shape, r, nslabs) for r in range(nslabs)]
    for s in slabs:
        sl = slice(s.plane0, s.plane1)
        s.set_boundary("difference_exponential", img[sl], 15.0, False); s.set_markers(fg[sl], bg[sl])
    ex = LoopbackExchange(slabs)
    for rep in range(2):
        t0 = time.perf_counter()
        for s in slabs: s.build()
        st = solve_slabs(slabs, ex)
        fl = sum(s.finish_device() for s in slabs)
        dt = time.perf_counter() - t0
        print(json.dumps({"shape": list(img.shape), "path": "%d slab(s), mgc_solve_slabs, all slabs on this GPU" % nslabs, "ms": round(dt * 1e3, 1), "mvox_s": round(img.size / dt / 1e6, 1), "flow": fl, **st}), flush=True)
    for s in slabs: s.close()
