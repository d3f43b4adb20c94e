"""SURVEY 8 row a15 (types / id widths): a voxel lattice with MORE THAN 2^31 voxels on one MI355X.

    python tools/gpu_big_ids.py            (needs ~215 GB of HBM and ~12 GB of host memory; about a minute)

The reference caps node ids at 2^31 (lib/maxflow/src/graph.h:57-62: node_id = int); here voxel ids are 64-bit and
tile ids 32-bit (medpy_amd/csrc/mgc_common.h).  No CPU oracle reaches this size, so the check is by construction: the
volume is THREE identical 768 x 1024 x 1024 blocks stacked along axis 0 (2 415 919 104 voxels), every block walled in by
background markers, so the three cuts are independent and must be identical -- the third block lives entirely above
voxel id 2^31.  Plus: the device-side invariants of a maximum preflow over the whole volume (mgc_validate), and
what_segment() on ids beyond 2^31 against the bulk label array.
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from medpy_amd import _lib  # noqa: E402
from medpy_amd.graphcut.graph import VoxelGraph  # noqa: E402

B, N, REP = 768, 1024, 3


def block():
    rng = np.random.default_rng(7)
    img = np.empty((B, N, N), np.uint8)
    fg = np.zeros((B, N, N), np.uint8)
    yy = (np.arange(N, dtype=np.float32) - (N - 1) / 2.0) ** 2
    r2_yx = yy[:, None] + yy[None, :]
    for z in range(B):
        r2 = r2_yx + np.float32((z - (B - 1) / 2.0) ** 2)
        img[z] = (100 * (r2 < 300.0 ** 2)).astype(np.uint8) + rng.integers(0, 40, (N, N), dtype=np.uint8)
        fg[z] = r2 < 80.0 ** 2
    bg = np.zeros((B, N, N), np.uint8)
    bg[0] = bg[-1] = 1
    bg[:, 0, :] = bg[:, -1, :] = 1
    bg[:, :, 0] = bg[:, :, -1] = 1
    return img, fg, bg


def main():
    t0 = time.time()
    try:
        img, fg, bg = (np.concatenate([a] * REP, axis=0) for a in block())
    except MemoryError:
        print("out of host memory for the input arrays", flush=True)
        sys.exit(77)
    shape = img.shape
    nvox = int(np.prod(shape))
    assert nvox > 2 ** 31
    print("volume %s = %d voxels (2^31 = %d), host arrays ready after %.0f s" % (shape, nvox, 2 ** 31, time.time() - t0), flush=True)
    try:
        g = VoxelGraph(shape)
        g._set_boundary("difference_exponential", img, 15.0, False)
        g._set_markers(fg, bg)
    except _lib.MedpyHipError as e:
        if e.code == _lib.ERR_OOM:
            print("out of device memory for %d voxels: %s" % (nvox, e), flush=True)
            sys.exit(77)
        raise
    t1 = time.perf_counter()
    g._build()
    flow = g.maxflow()
    dt = time.perf_counter() - t1
    st = g.stats()
    lab = g.labels()
    ok_blocks = bool(np.array_equal(lab[:B], lab[B:2 * B]) and np.array_equal(lab[:B], lab[2 * B:]))
    v = g.validate()
    _lib.assert_valid(v)
    ids = [2 ** 31 - 1, 2 ** 31, 2 ** 31 + 12345, nvox - 1, nvox - N * N * (B // 2) - N * (N // 2) - N // 2]
    seg_ok = all((g.what_segment(i) == g.termtype.SINK) == (not lab.flat[i]) for i in ids)
    out = {"shape": list(shape), "voxels": nvox, "above_2_31": nvox - 2 ** 31, "build_plus_solve_ms": round(dt * 1e3, 1),
           "mvox_s": round(nvox / dt / 1e6, 1), "flow": flow, "foreground_voxels": int(lab.sum()), "device_bytes": st["device_bytes"],
           "three_blocks_identical": ok_blocks, "what_segment_beyond_2_31_matches_labels": seg_ok, "node_num": int(g.get_node_num()),
           "validation": v, "global_relabels": st["global_relabels"], "discharge_tiles": st["discharge_tiles"]}
    print(json.dumps(out), flush=True)
    assert ok_blocks and seg_ok and v["voxels"] == nvox and out["node_num"] == nvox
    assert flow == v["cut_capacity"] + v["flow_constant"] or abs(flow - v["cut_capacity"] - v["flow_constant"]) <= 1e-9 * abs(flow)


if __name__ == "__main__":
    main()
