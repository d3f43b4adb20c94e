"""N slabs of the multi-GPU workload family time-multiplexed on ONE MI355X (development aid; profiles/r4_slab_scaling_one_gpu.jsonl):
what does cutting a volume into N slabs cost in kernel time, relabel passes, tile visits, exchanges and reductions?

    python tools/gpu_slab_scaling.py PLANES_PER_SLAB XY CONN N [N ...]       N = 1: the same volume on one handle (mgc_maxflow)

The volume is (PLANES_PER_SLAB * max(N)) x XY x XY (bench.block_volume: a grid of 512^3 sphere blocks sharing one medium), so
every N cuts the SAME volume; one process per N (a handle of this size created and destroyed before leaves the next one slower).
Per line: wall ms (the slabs take turns on one device: NOT a multi-GPU number), the kernels' own time summed per slab (HIP events
around the discharge / relabel launches; a slab's kernels run alone on the device, so this is what a device of its own would take),
work counters, exchanges / reductions, and the SHA-256 of the packed label volume.  SLAB_PARAMS="name=value,..." sets schedule
parameters of solve_slabs (exchange_every, exchange_rounds, radial ...)."""
import hashlib
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import bench  # noqa: E402

planes, xy, conn = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ns = [int(v) for v in sys.argv[4:]]
total = int(os.environ.get("SLAB_TOTAL_PLANES", planes * max(ns)))
if len(ns) > 1:
    for n in ns:
        subprocess.run([sys.executable, os.path.abspath(__file__), str(planes), str(xy), str(conn), str(n)], check=False,
                       env=dict(os.environ, SLAB_TOTAL_PLANES=str(total)))
    sys.exit(0)
n = ns[0]
block = 512 if xy % 512 == 0 and total % 512 == 0 else min(xy, 256)
img, fg, bg = bench.block_volume(0, total, total // block, xy // block, block)


def sha(lab):
    return hashlib.sha256(np.packbits(np.asarray(lab, dtype=np.uint8).ravel()).tobytes()).hexdigest()


if n == 1:
    from medpy_amd.graphcut.graph import VoxelGraph
    g = VoxelGraph(img.shape, connectivity=conn)
    g._set_boundary("difference_exponential", img, 15.0, False)
    g._set_markers(fg, bg)
    for kv in os.environ.get("SLAB_PARAMS", "").split(","):  # (the same knobs as the slab path below: A/Bs on one box)
        if kv:
            g.set_param(kv.split("=")[0], int(kv.split("=")[1]))
    best = None
    for rep in range(2):
        t0 = time.perf_counter(); g._build(); f = g.maxflow(); dt = time.perf_counter() - t0
        st = g.stats()
        rec = {"shape": list(img.shape), "conn": conn, "slabs": 1, "path": "single handle (mgc_maxflow)", "wall_ms": round(dt * 1e3, 1), "flow": f,
               "kernel_ms_per_slab": [round(st["build_ms"] + st["discharge_ms"] + st["relabel_ms"], 1)],
               **{k: (round(st[k], 1) if isinstance(st[k], float) else st[k]) for k in ("build_ms", "discharge_ms", "relabel_ms", "global_relabels", "phases", "discharge_tiles", "relabel_tiles", "relabel_launches", "readbacks")}}
        best = rec if best is None or rec["wall_ms"] < best["wall_ms"] else best
    best["labels_sha256"] = sha(g.labels())
    print(json.dumps(best), flush=True)
    g.close()
else:
    from medpy_amd.slab import HipSlab, LoopbackExchange, solve_slabs
    slabs = [HipSlab(img.shape, r, n, connectivity=conn) for r in range(n)]
    for s in slabs:
        sl = slice(s.plane0, s.plane1)
        s.set_boundary("difference_exponential", img[sl], 15.0, False); s.set_markers(fg[sl], bg[sl])

    ex = LoopbackExchange(slabs)
    best = None
    for rep in range(2):
        t0 = time.perf_counter()
        for s in slabs: s.build()
        kw = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("SLAB_PARAMS", "").split(",") if kv)}
        for k in [k for k in kw if k not in solve_slabs.__code__.co_varnames]:  # (a knob of the handles, not of the schedule: every slab gets it)
            for s in slabs: s.set_param(k, kw[k])
            del kw[k]
        st = solve_slabs(slabs, ex, **kw)
        parts = [s.finish() for s in slabs]
        dt = time.perf_counter() - t0
        sts = [s.stats() for s in slabs]
        rec = {"shape": list(img.shape), "conn": conn, "slabs": n, "path": "%d slabs time-multiplexed on one GPU, mgc_solve_slabs (the library's schedule; borders between the slabs' buffers in HBM)" % n, "wall_ms": round(dt * 1e3, 1),
               "flow": sum(p[1] for p in parts),
               "kernel_ms_per_slab": [round(q["build_ms"] + q["discharge_ms"] + q["relabel_ms"], 1) for q in sts],
               "discharge_ms_per_slab": [round(q["discharge_ms"], 1) for q in sts], "relabel_ms_per_slab": [round(q["relabel_ms"], 1) for q in sts],
               **st}
        if best is None or rec["wall_ms"] < best["wall_ms"]:
            best = rec
            best["labels_sha256"] = sha(np.concatenate([p[0] for p in parts], axis=0))
    print(json.dumps(best), flush=True)
    for s in slabs: s.close()
