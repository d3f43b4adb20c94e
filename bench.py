#!/usr/bin/env python3
"""bench.py -- Mvoxels/s of the voxel graph cut (build + solve) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 512]

A "step" = one pass of the hot path over one synthetic volume: n-link/t-link construction
(mgc_build) + max-flow solve (mgc_maxflow) with image and markers already resident in HBM and
the label array left in HBM (SURVEY.md 8(d) "headline, device-resident").  Workload = the
configuration BASELINE.json's metric is quoted on: 512^3 "sphere" volume, 6-connectivity,
boundary_difference_exponential, sigma 15.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, HIP-event timed on the launch
stream inside the library) and `cpu_baseline` (the reference's own BK solver, compiled in place
as oracle/_ref, timed on a bounded sample on the host cores).

N > 1: one process per GPU (torch.distributed launch contract, backend nccl = RCCL).  ONE volume of
(size*N, size, size) voxels -- N sphere blocks stacked along axis 0 -- is cut as N exact Z-slabs, one
per GPU (medpy_amd/slab.py): after every relabel pass / colour phase the packed slab borders
(labels + outbox flow) travel to the neighbour ranks with RCCL send/recv over xGMI and tiny
all-reduces decide termination.  Per-GPU work is fixed as N grows: weak scaling.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG_6CONN = 71.0  # algorithmic bytes per voxel, SURVEY.md 8(d): 4 + 2 + 2*(3*8 + 8) + 1
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic_per_launch():
    """HBM bytes per k_discharge launch from the committed rocprofv3 PMC passes (profiles/pmc_discharge.json, written by
    tools/rocpd_summary.py from `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this very command).
    FETCH_SIZE is doubled as MI355X_MICROARCH.md (HBM section) prescribes for wide coalesced reads on gfx950."""
    path = os.path.join(ROOT, "profiles", "pmc_discharge.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    return int((2.0 * d["fetch_kib_per_launch"] + d["write_kib_per_launch"]) * 1024)


def cpu_baseline(sample_n):
    """Reference BK (oracle/_ref, or the C restatement when it did not travel) on a bounded sample."""
    from medpy_amd import synthetic
    from oracle import bk, energy_numpy, pipeline
    s = synthetic.sphere((sample_n,) * 3)
    kind = bk.best_kind()
    w = energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"])  # NumPy part of the reference (not timed:
    # the reference spends its build time in the per-edge insertion, which IS timed below)
    best = None
    for _ in range(2):
        t0 = time.perf_counter()
        g = pipeline.build_graph(s["fg"], s["bg"], weights=w, kind=kind)
        t1 = time.perf_counter()
        g.maxflow()
        t2 = time.perf_counter()
        del g
        if best is None or (t2 - t0) < best[0]:
            best = (t2 - t0, t1 - t0, t2 - t1)
    n = sample_n ** 3
    return {
        "value": round(n / best[0] / 1e6, 4), "unit": "Mvoxels/s", "cores": 1,
        "kind": "reference" if kind == "ref" else "port",
        "sample": "%d^3 sphere volume, 6-conn, diff_exp sigma 15; bulk sum_edge build %.2fs + BK maxflow %.2fs, single thread "
                  "(the reference is single-threaded), host has %d cores" % (sample_n, best[1], best[2], os.cpu_count()),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--cpu-sample", type=int, default=320)
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))

    from medpy_amd import _lib, synthetic
    from medpy_amd.graphcut.graph import VoxelGraph

    if _lib.device_count() < 1:
        raise SystemExit("bench.py: no MI355X visible (the HIP path has no CPU fallback)")

    n = args.size
    acc = {"build_ms": 0.0, "solve_ms": 0.0, "discharge_ms": 0.0, "relabel_ms": 0.0, "discharge_launches": 0,
           "relabel_launches": 0, "discharge_tiles": 0, "relabel_tiles": 0, "global_relabels": 0, "phases": 0}
    flow = 0.0
    slab_stats = None
    rccl_stalled = False
    if world == 1:
        shape = (n, n, n)
        s = synthetic.sphere(shape, seed=0)
        g = VoxelGraph(shape, device=0)
        g._set_boundary("difference_exponential", s["image"], s["sigma"], False)  # H2D, outside the timed region
        g._set_markers(s["fg"], s["bg"])

        def step():
            g._build()
            return g.maxflow()

        for _ in range(args.warmup):
            step()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            flow = step()  # synchronous: returns after the stream drained
            st = g.stats()
            for k in acc:
                acc[k] += st[k]
        elapsed = time.perf_counter() - t0
        fg_fraction = float(g.labels().mean())
    else:
        import torch.distributed as dist  # out-of-band channel only (gloo): RCCL id broadcast, barriers, host scalars
        from medpy_amd.slab import DistExchange, HipSlab, RcclExchange, solve_slabs
        # MEDPY_DIST_BACKEND=gloo: development aid -- the borders travel through host buffers and the ranks may share
        # a GPU (exercises this code path on a 1-GPU box).  Default: RCCL over xGMI, driven by the library itself.
        backend = os.environ.get("MEDPY_DIST_BACKEND", "nccl")
        dev_index = local_rank % _lib.device_count()
        dist.init_process_group("gloo", rank=rank, world_size=world)
        gshape = (n * world, n, n)
        slab = HipSlab(gshape, rank, world, device=dev_index)
        # the local planes of the global volume: sphere block `b` occupies planes [b*n, (b+1)*n)
        b0, b1 = slab.plane0 // n, (slab.plane1 - 1) // n
        imgs, fgs, bgs = [], [], []
        for b in range(b0, b1 + 1):
            blk = synthetic.sphere((n, n, n), seed=b)
            bg = blk["bg"].copy()
            if b > 0:
                bg[0, 1:-1, 1:-1] = False  # interior block faces are not background: one connected medium
            if b < world - 1:
                bg[-1, 1:-1, 1:-1] = False
            imgs.append(blk["image"]); fgs.append(blk["fg"]); bgs.append(bg)
        sl = slice(slab.plane0 - b0 * n, slab.plane1 - b0 * n)
        img_local = np.concatenate(imgs, axis=0)[sl]
        fg_local, bg_local = np.concatenate(fgs, axis=0)[sl], np.concatenate(bgs, axis=0)[sl]
        slab.set_boundary("difference_exponential", img_local, 15.0, False)
        slab.set_markers(fg_local, bg_local)
        del imgs, fgs, bgs
        ex, transport = None, "gloo, host-staged (MEDPY_DIST_BACKEND=gloo)"
        if backend == "nccl":
            # RCCL is the transport.  Its bring-up (ncclCommInitRank + one border exchange + one counter all-reduce) runs
            # under a watchdog and the ranks agree on the outcome over gloo: if it fails or stalls on any rank, every rank
            # falls back to moving the same border buffers through host memory and the JSON line says so.
            import threading
            import torch
            box = {}

            def bring_up():
                try:
                    e = RcclExchange(slab)
                    slab.build()
                    e.exchange(0, 1, 4)
                    e.global_counts()
                    box["ex"] = e
                except Exception as err:  # noqa: BLE001 -- reported below
                    box["err"] = repr(err)

            th = threading.Thread(target=bring_up, daemon=True)
            th.start()
            th.join(float(os.environ.get("MEDPY_RCCL_TIMEOUT", "240")))
            ok = torch.tensor([1 if "ex" in box else 0], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok[0]) == 1:
                ex, transport = box["ex"], "RCCL (grouped ncclSend/ncclRecv between neighbour slabs)"
            else:
                hung = th.is_alive()
                sys.stderr.write("[bench rank %d] RCCL bring-up %s; falling back to host-staged borders\n" %
                                 (rank, "stalled" if hung else "failed: %s" % box.get("err", "on another rank")))
                transport = "gloo, host-staged (RCCL bring-up failed)"
                if hung:  # the handle is stuck inside the library on that thread: take a fresh one
                    rccl_stalled = True
                    slab = HipSlab(gshape, rank, world, device=dev_index)
                    slab.set_boundary("difference_exponential", img_local, 15.0, False)
                    slab.set_markers(fg_local, bg_local)
        if ex is None:
            ex = DistExchange(slab)

        def step():
            slab.build()
            st = solve_slabs([slab], ex)
            return st, slab.finish_device()

        for _ in range(args.warmup):
            step()
        dist.barrier()  # every library call above returned after its stream drained (device synchronised)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            slab_stats, part = step()  # finish_device() synchronises the stream
        dist.barrier()
        elapsed = time.perf_counter() - t0
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
        flow = float(ex.allreduce_sum([part]))
        lab, _ = slab.finish()
        fg_fraction = float(ex.allreduce_sum([float(lab.sum())])) / float(np.prod(gshape))
        dist.barrier()

    if rank == 0:
        nvox = n ** 3
        ms_per_step = elapsed / args.steps * 1e3
        value = world * nvox / (elapsed / args.steps) / 1e6
        # dominant kernel: k_discharge (tile region-discharge).  Units per launch = voxels of the tiles it visits.
        launches = max(acc["discharge_launches"], 1)
        avg_ms = acc["discharge_ms"] / launches
        vox_per_launch = acc["discharge_tiles"] * 512.0 / launches
        achieved = (B_ALG_6CONN * vox_per_launch) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": "Mvoxels/s graph-cut (build+solve), 512^3 6-conn; fraction of HBM roofline",
            "value": round(value, 3), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d^3 sphere volume (float32), 6-conn, boundary_difference_exponential sigma=15, "
                                   "fg=inner ball, bg=6 faces" % n,
                       "parallelism": ("%d exact Z-slabs of one %dx%dx%d volume (%d stacked sphere blocks), halo exchange between neighbour slabs" %
                                       (world, n * world, n, n, world)) if world > 1 else "single GPU",
                       "transport": transport if world > 1 else None,
                       "fg_fraction": round(fg_fraction, 5), "flow": flow},
            "phases_ms": {"build": round(acc["build_ms"] / args.steps, 3), "solve": round(acc["solve_ms"] / args.steps, 3),
                          "discharge_kernels": round(acc["discharge_ms"] / args.steps, 3),
                          "relabel_kernels": round(acc["relabel_ms"] / args.steps, 3),
                          "global_relabels": acc["global_relabels"] / args.steps, "colour_phases": acc["phases"] / args.steps,
                          "tile_discharges": acc["discharge_tiles"] / args.steps, "tile_relabels": acc["relabel_tiles"] / args.steps},
            "job_roofline_frac": round(value * 1e6 / world * B_ALG_6CONN / (HBM_PEAK_GBS * 1e9), 6),
            "roofline": {"bound": "hbm", "kernel": "k_discharge", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": pmc_traffic_per_launch(),
                         "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches / args.steps,
                         "voxels_per_launch": round(vox_per_launch, 1), "bytes_per_voxel": B_ALG_6CONN},
        }
        if slab_stats is not None:
            out["slab_schedule"] = slab_stats
            out["roofline"] = None  # per-kernel event timing is a single-GPU measurement (N=1 line)
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        elif not args.no_cpu:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()
        if rccl_stalled:  # a watchdog thread is still parked inside RCCL: skip interpreter teardown
            sys.stdout.flush()
            os._exit(0)


if __name__ == "__main__":
    main()
