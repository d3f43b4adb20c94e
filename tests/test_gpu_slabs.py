"""-m gpu: the Z-slab (multi-GPU) code path on ONE MI355X: N slab handles time-multiplexed on the
device, borders exchanged through the loopback transport (host buffers and HBM buffers).  Labels
must equal the single-handle labels and the oracle's, bit for bit."""
import numpy as np
import pytest

from oracle import pipeline

pytestmark = pytest.mark.gpu


def _single(s):
    from medpy_amd import graphcut
    g = graphcut.graph_from_voxels(s["fg"], s["bg"], boundary_term=graphcut.energy_voxel.boundary_difference_exponential,
                                   boundary_term_args=(s["image"], s["sigma"], False))
    flow = g.maxflow()
    return g.labels(), flow


@pytest.mark.parametrize("gen,shape,nslabs", [("sphere", (64, 48, 40), 2), ("sphere", (64, 48, 40), 4), ("hard", (48, 48, 48), 3),
                                              ("sphere", (37, 40, 24), 2), ("sphere", (128, 64, 64), 8)])
def test_slabs_equal_single_and_oracle(gen, shape, nslabs):
    from medpy_amd import synthetic
    from medpy_amd.slab import graphcut_voxel_slabs
    s = getattr(synthetic, gen)(shape)
    labels, flow, st = graphcut_voxel_slabs(s["image"], s["fg"], s["bg"], s["term"], s["sigma"], nslabs=nslabs)
    assert st["converged"] == 1
    single, sflow = _single(s)
    np.testing.assert_array_equal(labels, single)
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term=s["term"], image=s["image"], sigma=s["sigma"])
    np.testing.assert_array_equal(labels, ref.labels)
    assert flow == pytest.approx(ref.flow, rel=1e-9)
    assert flow == pytest.approx(sflow, rel=1e-12)


@pytest.mark.parametrize("term", ["difference_linear", "maximum_linear"])
def test_linear_terms_use_the_range_of_the_whole_volume(term):
    """energy_voxel.py:101 / 174-176 normalise by max |I| / |max - min| of the WHOLE image: the slabs reduce their local
    ranges first (sync_image_range), so every slab builds the capacities of the single-handle run; the intensity extremes
    sit in different slabs here, so a local range would give different weights on the two sides of every slab border"""
    from medpy_amd import graphcut, synthetic
    from medpy_amd.slab import graphcut_voxel_slabs
    shape = (48, 24, 32)
    s = synthetic.sphere(shape)
    img = s["image"].astype(np.float64)
    img[2, 3, 4], img[45, 20, 30] = -300.0, 700.0  # global min in the first slab, global max in the last
    fn = getattr(graphcut.energy_voxel, "boundary_" + term)
    g = graphcut.graph_from_voxels(s["fg"], s["bg"], boundary_term=fn, boundary_term_args=(img, False))
    flow = g.maxflow()
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term=term, image=img)
    assert flow == pytest.approx(ref.flow, rel=1e-9)
    for nslabs in (2, 3):
        labels, sflow, st = graphcut_voxel_slabs(img, s["fg"], s["bg"], term, None, nslabs=nslabs)
        assert st["converged"] == 1
        assert sflow == pytest.approx(ref.flow, rel=1e-9)
        np.testing.assert_array_equal(labels, g.labels())


def test_slab_handle_refuses_single_gpu_entry_points():
    from medpy_amd import _lib
    from medpy_amd.slab import HipSlab
    s = HipSlab((32, 16, 16), 0, 2)
    assert (s.own0, s.own1, s.plane0, s.plane1) == (0, 16, 0, 24) and s.has_hi and not s.has_lo
    s.set_boundary("difference_linear", np.zeros(s.local_shape, np.float32), None)
    with pytest.raises(_lib.MedpyHipError, match="WHOLE volume"):
        s.build()  # a slab cannot normalise a *_linear term by the range of its own planes
    s.set_image_range(0.0, 0.0, 0.0)
    s.build()
    import ctypes as C
    with pytest.raises(_lib.MedpyHipError):
        s._call("mgc_maxflow", C.byref(C.c_double()))
    s.close()


def test_rccl_transport_single_rank(tmp_path):
    """What can be checked of the native RCCL transport on a 1-GPU box: librccl is dlopen()ed, a 1-rank
    communicator initialises on the handle's device (the id travels through the package's own out-of-band channel, a directory of
    files -- no PyTorch in the process), ncclAllReduce of the counters round-trips, the exchange is a no-op without neighbours."""
    import sys
    from medpy_amd import synthetic
    from medpy_amd.rendezvous import FileStore
    from medpy_amd.slab import HipSlab, RcclExchange, solve_slabs
    import os
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    store = FileStore(0, 1, directory=str(tmp_path / "rdv"))
    shape = (40, 32, 32)
    s = synthetic.sphere(shape)
    slab = HipSlab(shape, 0, 1)
    slab.set_boundary(s["term"], s["image"], s["sigma"])
    slab.set_markers(s["fg"], s["bg"])
    slab.build()
    ex = RcclExchange(slab, store)
    st = solve_slabs([slab], ex)
    assert st["converged"] == 1
    labels, flow = slab.finish()
    single, sflow = _single(s)
    np.testing.assert_array_equal(labels, single)
    assert flow == pytest.approx(sflow, rel=1e-12)
    assert (slab.allreduce_counts() == slab.read_counts()).all()
    slab.close()
    store.close()


RCCL_WORKER = r'''
import hashlib, json, os, sys
import numpy as np
ROOT, out, rank, world, conn = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
sys.path.insert(0, ROOT)
from medpy_amd import synthetic
from medpy_amd.rendezvous import FileStore
from medpy_amd.slab import HipSlab, RcclExchange, solve_slabs, validate_slabs
shape = (128, 96, 96)
s = synthetic.sphere(shape)
store = FileStore(rank, world, directory=os.path.join(out, "rdv"), timeout=300)
slab = HipSlab(shape, rank, world, device=rank, connectivity=conn)
slab.set_boundary(s["term"], s["image"][slab.plane0:slab.plane1], s["sigma"], False)
slab.set_markers(s["fg"][slab.plane0:slab.plane1], s["bg"][slab.plane0:slab.plane1])
ex = RcclExchange(slab, store)   # rank 0's ncclUniqueId through the file store, ncclCommInitRank on every rank
slab.build()
st = solve_slabs([slab], ex)     # the library's own schedule: mgc_solve_slab, borders over ncclSend / ncclRecv
part = slab.finish_device()
flow = float(ex.allreduce_sum([part]))
lab, _ = slab.finish()
v = validate_slabs([slab], ex)
np.save(os.path.join(out, "labels_%d.npy" % rank), lab)
if rank == 0:
    json.dump({"flow": flow, "stats": st, "validation": v}, open(os.path.join(out, "result.json"), "w"))
slab.close()
store.close()
assert "torch" not in sys.modules
'''


@pytest.mark.parametrize("conn", [6, 26])
def test_two_ranks_over_real_rccl(tmp_path, conn):
    """TWO processes, TWO GPUs, the real librccl: the slab borders travel with ncclSend / ncclRecv, the counters with ncclAllReduce,
    the schedule is the library's own (mgc_solve_slab) -- and the assembled labels are those of the single handle and of the BK
    oracle.  Needs two devices: skipped on the 1-GPU boxes of the build pool, executed by whoever runs the GPU tier on a node."""
    import json
    import os
    import subprocess
    import sys
    from medpy_amd import _lib, synthetic
    from oracle import pipeline
    if _lib.device_count() < 2:
        pytest.skip("one GPU visible: real RCCL between two ranks needs two (mgc_device_count() = %d)" % _lib.device_count())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "worker.py"
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, NCCL_DEBUG="WARN", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), root, str(tmp_path), str(r), "2", str(conn)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    for p in procs:
        out, err = p.communicate(timeout=900)
        assert p.returncode == 0, (out[-2000:], err[-3000:])
    res = json.load(open(tmp_path / "result.json"))
    labels = np.concatenate([np.load(tmp_path / ("labels_%d.npy" % r)) for r in range(2)], axis=0)
    s = synthetic.sphere((128, 96, 96))
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term=s["term"], image=s["image"], sigma=s["sigma"], connectivity=conn)
    np.testing.assert_array_equal(labels, ref.labels)
    assert res["flow"] == pytest.approx(ref.flow, rel=1e-9)
    v = res["validation"]
    assert not any(v[k] for k in ("negative_values", "active_excess", "residual_arcs_across", "sink_links_across", "pair_violations", "node_violations", "pending_outbox"))


@pytest.mark.parametrize("gen,shape,nslabs,regional", [("sphere", (64, 40, 48), 2, False), ("hard", (48, 48, 40), 3, False),
                                                       ("sphere", (64, 48, 48), 4, True), ("sphere", (37, 40, 24), 2, False),
                                                       ("sphere", (128, 48, 48), 8, False)])
def test_full_neighbourhood_slabs_equal_single_and_oracle(gen, shape, nslabs, regional):
    """BASELINE config 5 in small: 26-neighbourhood (+ regional term) cut as Z-slabs; pushes over the slab border
    accumulate in the ghost tiles and travel in halo kind 1 (mgc26_halo_pack_tile)."""
    from medpy_amd import graphcut, synthetic
    from medpy_amd.slab import graphcut_voxel_slabs
    s = getattr(synthetic, gen)(shape)
    if regional:
        s.update(synthetic.regional(shape))
    reg = (s["prob"], s["alpha"]) if regional else None
    labels, flow, st = graphcut_voxel_slabs(s["image"], s["fg"], s["bg"], s["term"], s["sigma"], nslabs=nslabs, connectivity=26,
                                            regional=reg)
    assert st["converged"] == 1
    kw = dict(regional_term=graphcut.energy_voxel.regional_probability_map, regional_term_args=reg) if regional else {}
    g = graphcut.graph_from_voxels(s["fg"], s["bg"], boundary_term=graphcut.energy_voxel.boundary_difference_exponential,
                                   boundary_term_args=(s["image"], s["sigma"], False), connectivity=26, **kw)
    sflow = g.maxflow()
    np.testing.assert_array_equal(labels, g.labels())
    okw = dict(prob=s["prob"], alpha=s["alpha"]) if regional else {}
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term=s["term"], image=s["image"], sigma=s["sigma"], connectivity=26, **okw)
    np.testing.assert_array_equal(labels, ref.labels)
    assert flow == pytest.approx(ref.flow, rel=1e-9)
    assert flow == pytest.approx(sflow, rel=1e-9)


@pytest.mark.parametrize("halo_max", [0, 2])
@pytest.mark.parametrize("nranks,conn,gen,shape", [(2, 6, "sphere", (64, 40, 48)), (3, 6, "hard", (48, 48, 40)), (2, 26, "sphere", (64, 40, 48)),
                                                   (4, 26, "sphere", (64, 32, 32)),
                                                   (8, 6, "sphere", (128, 40, 48)), (8, 26, "sphere", (128, 32, 32))])  # (the rank count of configs 4 / 5's node)
def test_native_transport_protocol_with_mock_rccl(nranks, conn, gen, shape, halo_max, tmp_path, monkeypatch):
    """The multi-rank protocol of mgc_solve_slabs over the native channel (one grouped send / receive per neighbour and
    exchange, bounded compacted messages with deferral, the carry planes of the distance transforms as a pipeline of sends and
    receives, minimum and sum reductions, every rank taking the same decisions) -- with N ranks as N
    threads on ONE GPU over an in-process stand-in for librccl (tests/hostsim/mock_rccl.cpp: FIFO channels, a Send and
    its Recv must agree on the size).  Real RCCL refuses two ranks on one device; with it this path runs at
    bench.py --gpus N."""
    import json
    import os
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostsim"))
    import sim
    from medpy_amd import synthetic
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / "labels.npy")
    env = dict(os.environ, MEDPY_HIP_RCCL=sim.build_mock_rccl())
    if halo_max:  # two record slots per border message: nearly every exchange overflows, tiles are deferred and drained
        env["MEDPY_HIP_PARAMS"] = ",".join(filter(None, [env.get("MEDPY_HIP_PARAMS", ""), "halo_max_records=%d" % halo_max]))
    res = subprocess.run([sys.executable, os.path.join(root, "tests", "hostsim", "mock_rccl_worker.py"), root, str(nranks), str(conn), gen,
                          "x".join(str(v) for v in shape), out, "native"], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-2000:])
    info = json.loads(res.stdout.strip().splitlines()[-1])
    assert all(st["converged"] == 1 for st in info["stats"])
    s = getattr(synthetic, gen)(shape)
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term=s["term"], image=s["image"], sigma=s["sigma"], connectivity=conn)
    np.testing.assert_array_equal(np.load(out), ref.labels)
    assert info["flow"] == pytest.approx(ref.flow, rel=1e-9)


@pytest.mark.parametrize("flags,conn", [([], 6), (["--config", "5"], 26), (["--strong"], 6)])
def test_bench_multi_gpu_code_path_at_reduced_size(flags, conn):
    """bench.py --gpus 2 as the driver launches it (python -m torch.distributed.run, one process per rank: the launcher is the
    contract's, bench.py itself imports no PyTorch), at a reduced size and with the development transport (two ranks share the
    one GPU of this box, borders through host buffers and the file store): the workload
    generator (grid of sphere blocks, outer shell = background), the slab build, the distributed schedule, the
    device-side invariant check over all slabs that every N > 1 run ends with -- for the default (6-conn, config 4's
    family), --config 5 (26-conn) and --strong."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MEDPY_DIST_BACKEND="host", MEDPY_BENCH_ANY_WORLD="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29713", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--xy", "64",
           "--planes", "32", "--block", "32"] + flags
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert res.returncode == 0, (res.stdout[-3000:], res.stderr[-3000:])
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["connectivity"] == conn and out["config"]["shape"] == [64, 64, 64]
    assert out["scaling"] == ("strong" if "--strong" in flags else "weak")
    v = out["validation"]
    assert v["voxels"] == 64 ** 3 and not any(v[k] for k in ("negative_values", "active_excess", "residual_arcs_across",
                                                             "sink_links_across", "pair_violations", "node_violations", "pending_outbox"))
    assert "host-staged" in out["config"]["transport"]  # and the line says so: never mistaken for an RCCL number
    # the same volume on one handle gives the same cut
    sys.path.insert(0, root)
    import bench
    from medpy_amd import graphcut
    img, fg, bg = bench.block_volume(0, 64, 2, 2, 32)
    g = graphcut.graph_from_voxels(fg, bg, boundary_term=graphcut.energy_voxel.boundary_difference_exponential,
                                   boundary_term_args=(img, 15.0, False), **({"connectivity": 26} if conn == 26 else {}))
    assert out["config"]["flow"] == pytest.approx(g.maxflow(), rel=1e-9)
    assert out["config"]["fg_fraction"] == pytest.approx(float(g.labels().mean()), abs=1e-5)


def test_two_slabs_give_the_single_handle_labels_at_bench_size():
    """512 x 1024 x 1024 (the per-pair size of BASELINE config 4's family), 6-neighbourhood: the SHA-256 of the label volume two
    slabs assemble equals that of the same volume solved on one handle -- voxel for voxel, not only flow and invariants.
    Both fit one MI355X (46 GB each); a process per configuration (tools/gpu_slab_scaling.py)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    #  46 GB of HBM per configuration: skip, do not fail, on a device (or a host) that does not have them
    from medpy_amd import _lib
    free = _lib.device_free_bytes() if hasattr(_lib, "device_free_bytes") else None
    if free is not None and free < 56 * 2 ** 30:
        pytest.skip("needs ~46 GB of free HBM per configuration, %.0f GB free" % (free / 2 ** 30))
    recs = {}
    for n in (1, 2):
        res = subprocess.run([sys.executable, os.path.join(root, "tools", "gpu_slab_scaling.py"), "256", "1024", "6", str(n)],
                             env=dict(os.environ, SLAB_TOTAL_PLANES="512"), capture_output=True, text=True, timeout=1500, cwd=root)
        assert res.returncode == 0, (res.stdout[-2000:], res.stderr[-2000:])
        recs[n] = json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][-1])
    assert recs[1]["shape"] == recs[2]["shape"] == [512, 1024, 1024]
    assert recs[2]["converged"] == 1
    assert recs[2]["labels_sha256"] == recs[1]["labels_sha256"]
    assert recs[2]["flow"] == pytest.approx(recs[1]["flow"], rel=1e-12)
