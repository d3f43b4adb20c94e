#!/usr/bin/env python3
"""bench.py -- Mvoxels/s of the voxel graph cut (build + solve) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C] [--strong] [--no-cpu] [--no-extras] [--cpu-full]

A "step" = one pass of the hot path over one synthetic volume: n-link / t-link construction (mgc_build) + max-flow solve
with image and markers already resident in HBM and the label array left in HBM (SURVEY.md 8(d) "headline,
device-resident").

Workloads (BASELINE.json `configs`; shapes are (Z, Y, X)):

    N = 1  (default)        512^3 sphere volume, 6-neighbourhood, boundary_difference_exponential sigma 15: the headline
    N > 1  (default)        (256 N) x 1024 x 1024, 6-neighbourhood, one exact Z-slab of 256 planes per GPU (weak scaling):
                            N = 4 IS config 4 (1024^3, 6-conn, 4 GPUs), N = 8 has config 5's shape
    --config 2              256^3, 6-conn, one GPU            --config 3   512^3, 26-conn + regional_probability_map, one GPU
    --config 4              = --gpus 4 default                --config 5   2048 x 1024 x 1024, 26-conn, 8 slabs (needs --gpus 8)
    --strong                1024^3, 6-conn, cut into N slabs: the SAME volume at N = 1, 2, 4, 8 (strong scaling)

The multi-GPU volume is a grid of 512^3 sphere blocks sharing one connected medium (only the outer faces of the whole
volume are background); no CPU oracle reaches these sizes, so every N > 1 run ends with the device-side invariant check
(mgc_validate over all slabs: conservation, no residual arc across the cut, no active excess, flow == cut) and FAILS if it
does not hold.  After every relabel pass / colour phase the packed slab borders travel to the neighbour ranks with RCCL
send/recv over xGMI (medpy_amd/slab.py); tiny all-reduces decide termination.  The launcher of the build contract
(one process per GPU) only hands out RANK / LOCAL_RANK / WORLD_SIZE: bench.py and the package import no ML framework; the ranks meet in
a private directory of files (medpy_amd/rendezvous.py).

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel k_discharge_w, HIP-event timed on the launch stream inside
the library) and `cpu_baseline` (the reference's own BK solver, compiled in place as oracle/_ref, timed on a bounded sample
on the host cores; --cpu-full times the whole 512^3 instead, ~3 minutes and ~40 GB).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG = {6: 71.0, 26: 231.0}  # algorithmic bytes per voxel, SURVEY.md 8(d): 4 + 2 + 2*(ndir/2*8 + 8) + 1; + 4 with a probability map
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md
BLOCK = 512  # edge of one sphere block of the multi-GPU volume


def kernel_source_hash():
    """identifies the kernels a PMC pass was taken with (the GPU box has no git history)"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "medpy_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic_per_launch(name="pmc_discharge.json"):
    """HBM bytes per k_discharge_w launch from the rocprofv3 PMC passes of tools/profile_round.sh (FETCH_SIZE and WRITE_SIZE in
    separate passes over this very command; profiles/pmc_discharge.json; --config 3: profiles/pmc_discharge26.json, k26_discharge).  Only reported when those passes ran on the kernel
    sources of this tree (hash of medpy_amd/csrc); 8-byte-per-lane reads are not the access width FETCH_SIZE was calibrated
    for (MI355X_MICROARCH.md, HBM), so both the raw and the doubled read figure are given."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None
    d = json.load(open(path))
    info = {"fetch_kib_per_launch": d.get("fetch_kib_per_launch"), "write_kib_per_launch": d.get("write_kib_per_launch"),
            "kernel_sources": d.get("kernel_sources"), "matches_this_tree": d.get("kernel_sources") == kernel_source_hash()}
    if not info["matches_this_tree"] or d.get("fetch_kib_per_launch") is None or d.get("write_kib_per_launch") is None:
        return None, info
    return int((2.0 * d["fetch_kib_per_launch"] + d["write_kib_per_launch"]) * 1024), info


def host_memory_gb():
    try:
        for ln in open("/proc/meminfo"):
            if ln.startswith("MemAvailable:"):
                return int(ln.split()[1]) / 1048576.0
    except OSError:
        pass
    return 0.0


def cpu_baseline_in_run(sample_n, want_full):
    """The CPU leg of the line.  The whole 512^3 workload on the reference's own solver needs ~40 GB and ~5 minutes of one core: it
    runs (in a child process, so that a box without the memory or the time cannot take the GPU numbers down with it) when the
    host has >= 48 GB available, else the largest cube that fits -- 384^3 from 20 GB -- else the bounded sample."""
    import subprocess
    avail = host_memory_gb()
    n = BLOCK if (want_full and avail >= 48.0) else (384 if (want_full and avail >= 20.0) else sample_n)
    if n != sample_n:
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-only", str(n)], capture_output=True, text=True,
                                 timeout=float(os.environ.get("MEDPY_CPU_BASELINE_TIMEOUT", "900")))
            lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
            if res.returncode == 0 and lines:
                out = json.loads(lines[-1])
                out["host_memory_available_gb"] = round(avail, 1)
                return out
            sys.stderr.write("[bench] CPU baseline at %d^3 failed (rc %d): %s\n" % (n, res.returncode, res.stderr[-300:]))
        except subprocess.TimeoutExpired:
            sys.stderr.write("[bench] CPU baseline at %d^3 timed out; falling back to the %d^3 sample\n" % (n, sample_n))
    return cpu_baseline(sample_n)


def cpu_baseline(n):
    """Reference BK (oracle/_ref, or the C restatement when it did not travel) on an n^3 sphere volume: t_build = the
    reference's NumPy energies + the bulk sum_edge / add_tweights construction, t_solve = maxflow(); no label read-out, no
    hashing.  Plus the committed timing of the as-shipped Python path (BASELINE config 1) from the container
    that holds the reference."""
    from medpy_amd import synthetic
    from oracle import bk, energy_numpy, pipeline
    full = n == BLOCK
    s = synthetic.sphere((n,) * 3)
    kind = bk.best_kind()
    t0 = time.perf_counter()
    w = energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"])  # the reference's NumPy part (energy_voxel.py:611-664)
    tw = time.perf_counter() - t0
    best = None
    for _ in range(1 if full else 2):
        t0 = time.perf_counter()
        g = pipeline.build_graph(s["fg"], s["bg"], weights=w, kind=kind)
        t1 = time.perf_counter()
        g.maxflow()
        t2 = time.perf_counter()
        del g
        if best is None or (t2 - t0) < best[0]:
            best = (t2 - t0, t1 - t0, t2 - t1)
    out = {
        "value": round(n ** 3 / (best[0] + tw) / 1e6, 4), "unit": "Mvoxels/s", "cores": 1,
        "kind": "reference" if kind == "ref" else "port",
        "t_build_s": round(tw + best[1], 3), "t_solve_s": round(best[2], 3), "voxels": n ** 3,
        "sample": "%d^3 sphere volume (%s), 6-conn, diff_exp sigma 15; NumPy weights %.2fs + bulk sum_edge build %.2fs + BK maxflow %.2fs, "
                  "single thread (the reference is single-threaded), host has %d cores" % (
                      n, "the full workload" if full else "bounded sample of the 512^3 workload", tw, best[1], best[2], os.cpu_count()),
    }
    p = os.path.join(ROOT, "profiles", "cpu_python_path_64.json")
    if os.path.exists(p):  # BASELINE config 1: medpy_graphcut_voxel.py's own Python loop, measured where /root/reference exists
        out["python_path"] = json.load(open(p))
    p = os.path.join(ROOT, "profiles", "cpu_reference_512.json")
    if os.path.exists(p) and not full:
        out["full_workload_elsewhere"] = json.load(open(p))  # (measured in the build container, incl. read-out: for scale only)
    return out


def labels_sha256(labels):
    """the hash tests/golden/reference_large.json holds for the reference's label volumes (oracle/gen_golden.py large): SHA-256 of the packed bits"""
    return hashlib.sha256(np.packbits(np.asarray(labels).astype(np.uint8).ravel()).tobytes()).hexdigest()


def golden_large():
    p = os.path.join(ROOT, "tests", "golden", "reference_large.json")
    return json.load(open(p)) if os.path.exists(p) else {}


def parity_relaxation_summary():
    """the committed record of the last full GPU test tier (tests/conftest.py writes it on the GPU box): how many comparisons needed the
    equivalence checker (oracle/cutcheck.py: exact ties between minimum cuts) instead of strict label equality.  Historical -- values
    read from the file, nothing asserted about this run."""
    import glob
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_parity_relaxations.jsonl")))
    if not recs:
        return None
    rows = [json.loads(ln) for ln in open(recs[-1]) if ln.strip().startswith("{")]
    return {"file": "profiles/" + os.path.basename(recs[-1]), "relaxed_comparisons": len(rows), "most_voxels_differing": max([r.get("differing", 0) for r in rows] or [0]),
            "verdicts": sorted({str(r.get("verdict")) for r in rows}), "adjudicated_exactly": sum(1 for r in rows if r.get("exact_adjudication")),
            "note": "record of the last full GPU tier, not of this run"}


def also_case(name, n, conn, regional, steps=3, warmup=1, golden_key=None, workload="sphere"):
    """one more BASELINE config inside the driver's run: a fresh handle, `steps` timed build + solve steps (median), the dominant
    kernel's roofline fraction by the headline's formula, the device-side invariants, and the label hash against the
    reference's (tests/golden/reference_large.json) where the oracle reaches the size"""
    from medpy_amd import _lib, synthetic
    from medpy_amd.graphcut.graph import VoxelGraph
    shape = (n, n, n)
    s = getattr(synthetic, workload)(shape, seed=0)
    g = VoxelGraph(shape, device=0, connectivity=conn if conn != 6 else None)
    g._set_boundary("difference_exponential", s["image"], s["sigma"], False)
    g._set_markers(s["fg"], s["bg"])
    if regional:
        r = synthetic.regional(shape)
        g._set_regional(r["prob"], r["alpha"])
    times, flow, st = [], 0.0, None
    acc = {"discharge_ms": 0.0, "discharge_launches": 0, "discharge_tiles": 0, "discharge_wave_ms": 0.0, "discharge_wave_launches": 0, "discharge_wave_tiles": 0,
           "build_ms": 0.0, "relabel_ms": 0.0, "global_relabels": 0, "phases": 0}
    for i in range(warmup + steps):
        t0 = time.perf_counter()
        g._build()
        flow = g.maxflow()
        dt = time.perf_counter() - t0
        if i >= warmup:
            times.append(dt)
            st = g.stats()
            for k in acc:
                acc[k] += st[k]
    ms = float(np.median(times)) * 1e3
    if conn != 6:
        # the full neighbourhood's steps are ~100 launches of very different lengths: the kernel averages of this block come from ONE more
        # step (outside `times`) with an event pair around every launch instead of every 7th
        g.set_param("timing_stride", 1)
        g._build()
        g.maxflow()
        st = g.stats()
        acc = {k: st[k] * steps for k in acc}
    b_alg = B_ALG[conn] + (4.0 if regional else 0.0)
    wave = conn == 6 and acc["discharge_wave_launches"] > 0
    launches = max(acc["discharge_wave_launches"] if wave else acc["discharge_launches"], 1)
    avg_ms = (acc["discharge_wave_ms"] if wave else acc["discharge_ms"]) / launches
    vox = (acc["discharge_wave_tiles"] if wave else acc["discharge_tiles"]) * 512.0 / launches
    achieved = (b_alg * vox) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    v = g.validate()
    counts = ("negative_values", "active_excess", "residual_arcs_across", "sink_links_across", "pair_violations", "node_violations", "pending_outbox")
    out = {"workload": "%d^3 %s volume (%s), %d-conn%s" % (n, workload, s["image"].dtype.name, conn, " + regional_probability_map" if regional else ""),
           "ms_per_step": round(ms, 3), "mvoxels_s": round(n ** 3 / (ms * 1e-3) / 1e6, 1), "steps": steps, "flow": flow,
           "kernel": "k_discharge_w" if conn == 6 else ("k26_discharge_w" if regional else "k26_discharge"), "frac": round(achieved / HBM_PEAK_GBS, 5),
           "job_roofline_frac": round(n ** 3 / (ms * 1e-3) * b_alg / (HBM_PEAK_GBS * 1e9), 6),
           "build_ms": round(acc["build_ms"] / steps, 3), "discharge_kernels_ms": round(acc["discharge_ms"] / steps, 3), "relabel_kernels_ms": round(acc["relabel_ms"] / steps, 3),
           "global_relabels": acc["global_relabels"] / steps, "colour_phases": acc["phases"] / steps, "radial_cycles": st.get("radial_cycles"), "wall_tiles": st.get("wall_tiles"),
           "validation_all_zero": all(int(v[k]) == 0 for k in counts), "cut_capacity": v["cut_capacity"]}
    ref = golden_large().get(golden_key) if golden_key else None
    if ref is not None:
        sha = labels_sha256(g.labels())
        out["labels_sha256"] = sha
        out["labels_match_reference"] = bool(sha == ref["sha256_packed_labels"])
        out["reference"] = "tests/golden/reference_large.json:%s (unmodified reference BK, oracle/gen_golden.py large)" % golden_key
    g.close()
    return out


def _pool_info():
    from medpy_amd import _lib
    return _lib.pool_info(0)


def api_end_to_end(n=BLOCK, reps=5):
    """The public path from HOST arrays: graph_from_voxels(fg, bg, boundary_term, args) -> maxflow() -> labels(), wall clock per volume
    (H2D of image + markers, build, solve, read-out, D2H of the labels), median of `reps`; the reference's own call sequence
    (bin/medpy_graphcut_voxel.py:163-182) with the per-voxel what_segment loop replaced by the bulk read-out."""
    from medpy_amd import graphcut, synthetic
    s = synthetic.sphere((n, n, n), seed=0)
    times, parts, labels = [], [], None
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        g = graphcut.graph_from_voxels(s["fg"], s["bg"], boundary_term=graphcut.energy_voxel.boundary_difference_exponential,
                                       boundary_term_args=(s["image"], s["sigma"], False))
        t1 = time.perf_counter()
        flow = g.maxflow()
        t2 = time.perf_counter()
        labels = g.labels()
        t3 = time.perf_counter()
        g.close()
        t4 = time.perf_counter()
        times.append(t3 - t0)
        parts.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
    times, parts = times[1:], parts[1:]  # (the first call pays the handle's first allocations: every later one is served by the library's pool, mgc_pool_*)
    ms = float(np.median(times)) * 1e3
    slow = int(np.argmax(times))

    def split(p):
        return {"graph_from_voxels (handle + H2D + build)": round(p[0] * 1e3, 2), "maxflow": round(p[1] * 1e3, 2), "labels (read-out + D2H)": round(p[2] * 1e3, 2),
                "close (not in each_ms)": round(p[3] * 1e3, 2)}
    ref = golden_large().get("sphere_512_6") if n == 512 else None
    out = {"ms_per_volume": round(ms, 2), "mvoxels_s": round(n ** 3 / (ms * 1e-3) / 1e6, 1), "reps": reps, "each_ms": [round(t * 1e3, 2) for t in times],
           "last_call_ms": split(parts[-1]), "slowest_call_ms": split(parts[slow]), "device_memory_pool": _pool_info(),
           "flow": flow, "path": "medpy_amd.graphcut.graph_from_voxels(...).maxflow(); .labels() from host arrays (float32 image, bool markers)"}
    if ref is not None:
        out["labels_match_reference"] = bool(labels_sha256(labels) == ref["sha256_packed_labels"])
    return out


def block_volume(planes0, planes1, nz_blocks, xy_blocks, block):
    """the local planes [planes0, planes1) of a (nz_blocks x xy_blocks x xy_blocks) grid of sphere blocks: image, fg, bg.
    Every block carries its own bright ball and foreground seed; exactly the outer shell of the WHOLE volume is
    background, so the blocks share one connected medium."""
    from medpy_amd import synthetic
    b0, b1 = planes0 // block, (planes1 - 1) // block
    slabs_i, slabs_f = [], []
    for bz in range(b0, b1 + 1):
        z0, z1 = max(planes0, bz * block) - bz * block, min(planes1, (bz + 1) * block) - bz * block
        rows_i, rows_f = [], []
        for by in range(xy_blocks):
            ri, rf = [], []
            for bx in range(xy_blocks):
                blk = synthetic.sphere((block,) * 3, seed=(bz * xy_blocks + by) * xy_blocks + bx)
                ri.append(blk["image"][z0:z1]); rf.append(blk["fg"][z0:z1])
            rows_i.append(np.concatenate(ri, axis=2)); rows_f.append(np.concatenate(rf, axis=2))
        slabs_i.append(np.concatenate(rows_i, axis=1)); slabs_f.append(np.concatenate(rows_f, axis=1))
    img, fg = np.concatenate(slabs_i, axis=0), np.concatenate(slabs_f, axis=0)
    bg = np.zeros(img.shape, dtype=bool)
    if planes0 == 0:
        bg[0] = True
    if planes1 == nz_blocks * block:
        bg[-1] = True
    bg[:, 0, :] = True; bg[:, -1, :] = True; bg[:, :, 0] = True; bg[:, :, -1] = True
    return np.ascontiguousarray(img), np.ascontiguousarray(fg), bg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=7)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=0, help="BASELINE.json config 2..5 (default: by --gpus, see the module docstring)")
    ap.add_argument("--strong", action="store_true", help="1024^3, 6-conn, cut into --gpus slabs: strong scaling")
    ap.add_argument("--size", type=int, default=0, help="edge of the single-GPU cube (default 512; config 2: 256)")
    ap.add_argument("--xy", type=int, default=1024, help="cross-section edge of the multi-GPU volume (a multiple of --block)")
    ap.add_argument("--planes", type=int, default=256, help="planes per GPU of the multi-GPU volume")
    ap.add_argument("--block", type=int, default=BLOCK, help="edge of one sphere block of the multi-GPU volume")
    ap.add_argument("--cpu-sample", type=int, default=320)
    ap.add_argument("--cpu-full", action="store_true", help="(default when the host has the memory) the CPU leg runs the whole 512^3")
    ap.add_argument("--cpu-sample-only", action="store_true", help="CPU leg on the bounded sample whatever the host")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only: no `also` (configs 2 / 3) and no `api_end_to_end` block (profiling runs)")
    ap.add_argument("--cpu-only", type=int, default=0, help=argparse.SUPPRESS)  # child process of cpu_baseline_in_run
    args = ap.parse_args()
    if args.cpu_only:
        print(json.dumps(cpu_baseline(args.cpu_only)))
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))

    from medpy_amd import _lib, synthetic
    from medpy_amd.graphcut.graph import VoxelGraph

    ndev = _lib.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no MI355X visible (the HIP path has no CPU fallback)")

    conn, regional = 6, False
    if args.config == 3:
        conn, regional = 26, True
    if args.config == 5:
        conn = 26
        if world != 8 and not os.environ.get("MEDPY_BENCH_ANY_WORLD"):
            raise SystemExit("bench.py: --config 5 is 2048 x 1024 x 1024 on 8 GPUs (launch with --gpus 8)")
    if args.config == 4 and world != 4 and not os.environ.get("MEDPY_BENCH_ANY_WORLD"):
        raise SystemExit("bench.py: --config 4 is 1024^3 on 4 GPUs (launch with --gpus 4)")

    acc = {"build_ms": 0.0, "solve_ms": 0.0, "discharge_ms": 0.0, "relabel_ms": 0.0, "discharge_launches": 0,
           "relabel_launches": 0, "discharge_tiles": 0, "relabel_tiles": 0, "global_relabels": 0, "phases": 0,
           "discharge_wave_ms": 0.0, "discharge_wave_launches": 0, "discharge_wave_tiles": 0}
    flow = 0.0
    step_s = []  # wall time of every timed step (N > 1: the slowest rank's)
    end_to_end = None
    head_sha = None
    slab_stats = validation = None
    transport = None
    if world == 1 and not args.strong:
        n = args.size or (256 if args.config == 2 else BLOCK)
        shape = (n, n, n)
        s = synthetic.sphere(shape, seed=0)
        g = VoxelGraph(shape, device=0, connectivity=conn if conn != 6 else None)
        if regional:
            r = synthetic.regional(shape)
        t_h2d = time.perf_counter()
        g._set_boundary("difference_exponential", s["image"], s["sigma"], False)  # H2D (synchronous copies), outside the timed region
        g._set_markers(s["fg"], s["bg"])
        if regional:
            g._set_regional(r["prob"], r["alpha"])
        t_h2d = time.perf_counter() - t_h2d

        def step():
            g._build()
            return g.maxflow()

        for _ in range(args.warmup):
            step()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ts = time.perf_counter()
            flow = step()  # synchronous: returns after the stream drained
            step_s.append(time.perf_counter() - ts)
            st = g.stats()
            for k in acc:
                acc[k] += st[k]
        elapsed = time.perf_counter() - t0
        t_d2h = time.perf_counter()
        lab = g.labels()  # D2H of the label volume (one byte per voxel)
        t_d2h = time.perf_counter() - t_d2h
        fg_fraction = float(lab.mean())
        h2d_bytes = s["image"].nbytes + s["fg"].size + s["bg"].size + (r["prob"].nbytes if regional else 0)
        end_to_end = {"h2d_ms": round(t_h2d * 1e3, 2), "h2d_bytes": int(h2d_bytes), "d2h_ms": round(t_d2h * 1e3, 2), "d2h_bytes": int(lab.size),
                      "note": "image + markers (+ probability map) from pageable host arrays into HBM (pinned staging, four host threads), labels back; outside the timed steps; never part of `value`"}
        validation = g.validate()
        _lib.assert_valid(validation)
        t_h2d_cold = t_h2d  # (the first upload of the process also pins the staging buffers; the inputs go up once more, warm)
        t_h2d = time.perf_counter()
        g._set_boundary("difference_exponential", s["image"], s["sigma"], False)
        g._set_markers(s["fg"], s["bg"])
        if regional:
            g._set_regional(r["prob"], r["alpha"])
        t_h2d = time.perf_counter() - t_h2d
        end_to_end["h2d_ms"], end_to_end["h2d_first_call_ms"] = round(t_h2d * 1e3, 2), round(t_h2d_cold * 1e3, 2)
        head_sha = None
        ref = golden_large().get("sphere_512_6") if (n == 512 and conn == 6 and not regional) else (golden_large().get("sphere_256_6") if (n == 256 and conn == 6 and not regional) else None)
        if ref is not None:
            head_sha = {"labels_sha256": labels_sha256(lab), "reference": ref["sha256_packed_labels"]}
            head_sha["labels_match_reference"] = bool(head_sha["labels_sha256"] == head_sha["reference"])
        g.close()
        gshape = shape
        workload = "%d^3 sphere volume (float32), %d-conn, boundary_difference_exponential sigma=15%s, fg=inner ball, bg=6 faces" % (
            n, conn, " + regional_probability_map (float32, alpha 0.5)" if regional else "")
    else:
        from medpy_amd.rendezvous import FileStore
        from medpy_amd.slab import HipSlab, LoopbackExchange, RcclExchange, StoreExchange, solve_slabs, sync_boundary_table, validate_slabs
        # The launcher of the contract only provides RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; nothing here imports an ML framework.  Out-of-band channel (the RCCL id, barriers, a few host scalars): a private directory of files
        # (medpy_amd/rendezvous.py).  MEDPY_DIST_BACKEND=host: development aid -- the borders travel through host buffers and that
        # directory, the ranks may share a GPU (exercises this code path on a 1-GPU box).  Default: RCCL over xGMI, driven by the
        # library itself; if it cannot be brought up, or there are fewer GPUs than ranks, the run FAILS rather than print a
        # number that is not an RCCL number.
        backend = os.environ.get("MEDPY_DIST_BACKEND", "nccl")
        if backend == "gloo":
            backend = "host"  # (the name rounds 1 - 4 used for the development transport)
        if world > ndev and backend == "nccl":
            raise SystemExit("bench.py: %d ranks but %d GPUs visible (MEDPY_DIST_BACKEND=host shares GPUs for development runs)" % (world, ndev))
        dev_index = local_rank % ndev
        store = FileStore(rank, world, timeout=float(os.environ.get("MEDPY_RENDEZVOUS_TIMEOUT", "900"))) if world > 1 else None
        xy, blk = args.xy, args.block
        if xy % blk:
            raise SystemExit("bench.py: --xy must be a multiple of --block")
        Z = xy if args.strong else args.planes * world  # strong: one cube for every N; weak: --planes per GPU
        if Z % blk:
            raise SystemExit("bench.py: %d planes are not a whole number of %d-plane blocks" % (Z, blk))
        gshape = (Z, xy, xy)
        slab = HipSlab(gshape, rank, world, device=dev_index, connectivity=conn)
        img_local, fg_local, bg_local = block_volume(slab.plane0, slab.plane1, Z // blk, xy // blk, blk)
        slab.set_boundary("difference_exponential", img_local, 15.0, False)
        slab.set_markers(fg_local, bg_local)
        del img_local, fg_local, bg_local
        if world == 1:
            ex, transport = LoopbackExchange([slab]), "none (one slab)"
        elif backend == "nccl":
            # bring-up (ncclCommInitRank + one border exchange + one counter all-reduce) under a watchdog: a stall must end the
            # run with an error, not hang the node and not fall back to a transport whose number would be taken for RCCL's
            import threading
            box = {}

            def bring_up():
                try:
                    e = RcclExchange(slab, store)
                    slab.build()
                    slab.exchange(0, 1, 4)   # (one border message each way and one counter reduction: the channel works before the clock starts)
                    slab.allreduce_counts()
                    box["ex"] = e
                except Exception as err:  # noqa: BLE001 -- reported below
                    box["err"] = repr(err)

            th = threading.Thread(target=bring_up, daemon=True)
            th.start()
            th.join(float(os.environ.get("MEDPY_RCCL_TIMEOUT", "300")))
            # (the other ranks' verdict comes through files of their own names: a rank stalled inside RCCL does not block this)
            ok = store.allreduce([1.0 if "ex" in box else 0.0], "min")
            if int(ok[0]) != 1:
                sys.stderr.write("[bench rank %d] RCCL bring-up %s -- no result\n" % (rank, "stalled" if th.is_alive() else "failed: %s" % box.get("err", "on another rank")))
                sys.stderr.flush()
                os._exit(3)
            ex, transport = box["ex"], "RCCL (grouped ncclSend/ncclRecv between neighbour slabs, ncclAllReduce of the counters)"
        else:
            ex, transport = StoreExchange(slab, store), "host-staged borders through files (MEDPY_DIST_BACKEND=host: development run, not an RCCL number)"

        # integer-valued images: the term by table is ONE decision for the whole volume; a float image leaves the device's exp
        sync_boundary_table([slab], ex if world > 1 else LoopbackExchange([slab]))
        wall = {"build": 0.0, "solve": 0.0}

        def step():
            t_a = time.perf_counter()
            slab.build()
            t_b = time.perf_counter()
            st = solve_slabs([slab], ex)
            if not st.get("converged", 1):  # (the library's schedule hands back the stats of a run that exhausted max_outer)
                raise SystemExit("bench.py: the slab schedule did not reach a maximum preflow: %r" % (st,))
            part_flow = slab.finish_device()
            wall["build"], wall["solve"] = (t_b - t_a) * 1e3, (time.perf_counter() - t_b) * 1e3  # (the last step's, this rank's)
            return st, part_flow

        for _ in range(args.warmup):
            step()
        if world > 1:
            store.barrier()  # every library call above returned after its stream drained (device synchronised)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ts = time.perf_counter()
            slab_stats, part = step()  # finish_device() synchronises the stream
            step_s.append(time.perf_counter() - ts)
        if world > 1:
            store.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = store.allreduce([elapsed] + step_s, "max")  # a step ends when its slowest rank is done (every exchange synchronises neighbours)
            elapsed, step_s = float(t[0]), [float(v) for v in t[1:]]
        flow = float(ex.allreduce_sum([part]))
        lab, _ = slab.finish()
        fg_fraction = float(ex.allreduce_sum([float(lab.sum())])) / float(np.prod(gshape))
        validation = validate_slabs([slab], ex)  # no oracle reaches this size: the invariants are the check
        _lib.assert_valid(validation)
        assert abs((validation["cut_capacity"] + validation["flow_constant"]) - flow) <= 1e-9 * max(abs(flow), 1e-300)
        if world > 1:
            store.barrier()
        workload = "%dx%dx%d volume (%dx%dx%d sphere blocks of %d^3, float32, one connected medium), %d-conn, boundary_difference_exponential sigma=15, bg=outer faces" % (
            gshape[0], gshape[1], gshape[2], gshape[0] // blk, xy // blk, xy // blk, blk, conn)

    if rank == 0:
        nvox = float(np.prod(gshape))
        mean_ms = elapsed / args.steps * 1e3
        ms_per_step = float(np.median(step_s)) * 1e3 if step_s else mean_ms  # "median of >= 5"; the mean of the K steps stands beside it
        value = nvox / (ms_per_step * 1e-3) / 1e6
        b_alg = B_ALG[conn] + (4.0 if regional else 0.0)
        out = {
            # BASELINE.json's metric, quoted on config 2 (512^3, 6-conn); other configs name their own shape and neighbourhood
            "metric": "Mvoxels/s graph-cut (build+solve), %s %d-conn; fraction of HBM roofline" % (
                "512^3" if tuple(gshape) == (512, 512, 512) else "x".join(str(v) for v in gshape), conn),
            "value": round(value, 3), "unit": "Mvoxels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "mean_ms_per_step": round(mean_ms, 3), "step_ms": [round(v * 1e3, 3) for v in step_s],
            "value_is": "voxels / median step time (inputs resident in HBM, labels left in HBM)", "higher_is_better": True, "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "shape": list(gshape), "connectivity": conn,
                       "baseline_config": args.config or (None if args.strong else {1: "headline", 4: 4}.get(world)),
                       "parallelism": ("%d exact Z-slabs (one per GPU, %d planes each), halo exchange between neighbour slabs" % (
                           world, gshape[0] // world)) if world > 1 else "single GPU",
                       "transport": transport, "fg_fraction": round(fg_fraction, 5), "flow": flow},
            "job_roofline_frac": round(value * 1e6 / world * b_alg / (HBM_PEAK_GBS * 1e9), 6),
            "per_gpu_algorithmic_gbs": round(value * 1e6 / world * b_alg / 1e9, 2),
            "validation": validation,
        }
        if end_to_end is not None:
            end_to_end["ms_per_volume"] = round(ms_per_step + end_to_end["h2d_ms"] + end_to_end["d2h_ms"], 3)
            end_to_end["mvoxels_s"] = round(nvox / (end_to_end["ms_per_volume"] * 1e-3) / 1e6, 1)
            out["end_to_end"] = end_to_end
        if slab_stats is None:
            # dominant kernel: k_discharge_w (region discharge, one wave per tile).  Units per launch = voxels of the tiles it visits.
            # (the short lists of a solve go to the workgroup-per-tile kernel k_discharge: its launches, time and tiles are NOT in here)
            wave = conn == 6 and acc["discharge_wave_launches"] > 0
            launches = max(acc["discharge_wave_launches"] if wave else acc["discharge_launches"], 1)
            avg_ms = (acc["discharge_wave_ms"] if wave else acc["discharge_ms"]) / launches
            vox_per_launch = (acc["discharge_wave_tiles"] if wave else acc["discharge_tiles"]) * 512.0 / launches
            achieved = (b_alg * vox_per_launch) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            traffic, traffic_info = (pmc_traffic_per_launch() if conn == 6 and not args.config else
                                     (pmc_traffic_per_launch("pmc_discharge26.json") if args.config == 3 else (None, None)))
            out["phases_ms"] = {"build": round(acc["build_ms"] / args.steps, 3), "solve": round(acc["solve_ms"] / args.steps, 3),
                                "discharge_kernels": round(acc["discharge_ms"] / args.steps, 3),
                                "relabel_kernels": round(acc["relabel_ms"] / args.steps, 3),
                                "global_relabels": acc["global_relabels"] / args.steps, "colour_phases": acc["phases"] / args.steps,
                                "tile_discharges": acc["discharge_tiles"] / args.steps, "tile_relabels": acc["relabel_tiles"] / args.steps}
            # (a graph with a regional term is pre-pushed by k_build and discharged by the one-wave-per-tile kernel, mgc_maxflow's choice)
            out["roofline"] = {"bound": "hbm", "kernel": "k_discharge_w" if conn == 6 else ("k26_discharge_w" if regional else "k26_discharge"), "achieved": round(achieved, 2),
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                               "traffic_source": traffic_info, "avg_launch_ms": round(avg_ms, 4), "launches_per_step": launches / args.steps,
                               "voxels_per_launch": round(vox_per_launch, 1), "bytes_per_voxel": b_alg,
                               "timing": "HIP event pairs on the launch stream around every %d-th launch of the kernel (the timed residue rotates from step to step), mean x launches" % max(int(st.get("timing_stride", 1)), 1),
                               "other_discharge_launches_per_step": (acc["discharge_launches"] - launches) / args.steps if wave else 0.0}
        else:
            out["slab_schedule"] = slab_stats
            # the dominant kernel on THIS rank's slab (rank 0; the library times its own launches in the slab driver too)
            pst = slab.stats()
            wave = conn == 6 and pst.get("discharge_wave_launches", 0) > 0
            launches = max(pst["discharge_wave_launches"] if wave else pst["discharge_launches"], 1)
            avg_ms = (pst["discharge_wave_ms"] if wave else pst["discharge_ms"]) / launches
            vox_per_launch = (pst["discharge_wave_tiles"] if wave else pst["discharge_tiles"]) * 512.0 / launches
            achieved = (b_alg * vox_per_launch) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            out["phases_ms"] = {"build": round(wall["build"], 3), "solve": round(wall["solve"], 3), "discharge_kernels": round(pst["discharge_ms"], 3),
                                "relabel_kernels": round(pst["relabel_ms"], 3), "rank": 0}
            timed = avg_ms > 0  # the library's own schedule (mgc_solve_slab, RCCL transport) times its launches; a schedule driven from Python does not
            out["roofline"] = {"bound": "hbm", "kernel": "k_discharge_w" if conn == 6 else "k26_discharge", "achieved": round(achieved, 2) if timed else None,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5) if timed else None, "traffic": None,
                               "avg_launch_ms": round(avg_ms, 4) if timed else None, "launches_per_step": launches if timed else None,
                               "voxels_per_launch": round(vox_per_launch, 1) if timed else None,
                               "bytes_per_voxel": b_alg, "scope": "rank 0's slab, last step; per-GPU algorithmic GB/s of the whole job: per_gpu_algorithmic_gbs"}
            if not timed:
                out["roofline"]["timing"] = ("not available in this run: the launches of a slab are timed by the library's own schedule (mgc_solve_slab over "
                                             "RCCL); this run drove the schedule from Python over the development transport")
        if world == 1 and not args.strong and head_sha is not None:
            out["config"].update(head_sha)
        out["parity"] = parity_relaxation_summary()
        if world == 1 and not args.config and not args.strong and not args.no_extras and not args.size:
            # the other single-GPU BASELINE configs and the public API path, inside the driver's run
            out["api_end_to_end"] = api_end_to_end()
            out["also"] = {"config2": also_case("config2", 256, 6, False, golden_key="sphere_256_6"),
                           "config3": also_case("config3", 512, 26, True),
                           "config3_at_256": also_case("config3_at_256", 256, 26, True, steps=1, warmup=1, golden_key="config3_256_26_regional"),
                           # the inputs the headline volume says nothing about: weak contrast, ties everywhere, the
                           # 26-neighbourhood without a regional term (config 5's per-GPU workload), an integer-valued (CT-like, uint16) volume
                           "hard": also_case("hard", 512, 6, False, steps=2, workload="hard"),
                           "ties": also_case("ties", 512, 6, False, steps=2, warmup=0, workload="ties"),
                           "conn26_markers": also_case("conn26_markers", 512, 26, False, steps=2, warmup=0),
                           "ct_uint16": also_case("ct_uint16", 512, 6, False, steps=2, workload="ct")}
        if not args.no_cpu and world == 1 and not args.config and not args.strong:
            out["cpu_baseline"] = cpu_baseline_in_run(args.cpu_sample, not args.cpu_sample_only)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        store.close()


if __name__ == "__main__":
    main()
