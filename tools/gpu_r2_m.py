import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd import synthetic
from medpy_amd.graphcut.graph import VoxelGraph
for n in (128, 256, 512):
    s = synthetic.sphere((n, n, n))
    g = VoxelGraph((n, n, n))
    g._set_boundary("difference_exponential", s["image"], s["sigma"], False)
    g._set_markers(s["fg"], s["bg"])
    g.set_param("kernel_timing", 0)
    for thr in (0, 256, 512, 1024, 2048, 1 << 30):
        g.set_param("wave_min_tiles", thr)
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); g._build(); g.maxflow(); best = min(best, time.perf_counter() - t0)
        st = g.stats()
        print(json.dumps({"n": n, "wave_min_tiles": thr, "ms": round(best * 1e3, 2), "dis_tiles": st["discharge_tiles"], "rel_tiles": st["relabel_tiles"], "relabels": st["global_relabels"]}), flush=True)
