"""the multi-GPU workload family on ONE handle: how hard is the volume itself, how much does the slab driver add? (development aid)"""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from medpy_amd.graphcut.graph import VoxelGraph
Z, XY = int(sys.argv[1]), int(sys.argv[2])
img, fg, bg = bench.block_volume(0, Z, Z // 512, XY // 512, 512)
g = VoxelGraph(img.shape)
g._set_boundary("difference_exponential", img, 15.0, False)
g._set_markers(fg, bg)
for rep in range(2):
    t0 = time.perf_counter(); g._build(); f = g.maxflow(); dt = time.perf_counter() - t0
    st = g.stats()
    print(json.dumps({"shape": list(img.shape), "path": "single handle, mgc_maxflow", "ms": round(dt * 1e3, 1), "mvox_s": round(img.size / dt / 1e6, 1), "flow": f,
                      **{k: (round(st[k], 1) if isinstance(st[k], float) else st[k]) for k in ("build_ms", "discharge_ms", "relabel_ms", "global_relabels", "phases", "discharge_tiles", "relabel_tiles", "readbacks")}}), flush=True)
g.close()
from medpy_amd.slab import HipSlab, LoopbackExchange, solve_slabs
for nslabs in (1, 2):
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
        print(json.dumps({"shape": list(img.shape), "path": "%d slab(s), Python schedule, loopback" % nslabs, "ms": round(dt * 1e3, 1), "mvox_s": round(img.size / dt / 1e6, 1), "flow": fl, **st}), flush=True)
    for s in slabs: s.close()
