"""CPU tier: the multi-GPU Z-slab protocol (medpy_amd/slab.py: schedule + transports) executed over
the host simulator.  Loopback (all slabs in-process) and a real 2-process gloo run must both give
exactly the labels of the BK oracle (= the single-slab labels)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))


def _problem(gen, shape):
    from medpy_amd import synthetic
    from oracle import energy_numpy, pipeline
    s = getattr(synthetic, gen)(shape)
    w = energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"])
    g = pipeline.build_graph(s["fg"], s["bg"], weights=w)
    tr = np.array([g.get_trcap(i) for i in range(s["fg"].size)])
    g.maxflow()
    return w, tr, g.labels().reshape(shape).astype(bool)


@pytest.mark.parametrize("gen,shape,nslabs", [("sphere", (32, 24, 24), 2), ("sphere", (40, 24, 16), 3), ("hard", (32, 32, 32), 4),
                                              ("sphere", (21, 16, 24), 2), ("sphere", (64, 16, 16), 8)])
def test_loopback_slabs_match_oracle(gen, shape, nslabs):
    import sim
    from medpy_amd.slab import LoopbackExchange, solve_slabs
    w, tr, ref = _problem(gen, shape)
    slabs = [sim.SimSlab(shape, r, nslabs) for r in range(nslabs)]
    for s in slabs:
        s.load(w, tr)
    st = solve_slabs(slabs, LoopbackExchange(slabs))
    assert st["converged"] == 1
    labels = np.concatenate([s.finish()[0] for s in slabs], axis=0)
    np.testing.assert_array_equal(labels, ref)
    # the borders travel once per round of the two colours, not after every phase: fewer exchanges than phases + relabel rounds
    assert st["exchanges"] < st["phases"] + st["relabel_passes"], st
    # slab boundaries fall on tile layers and partition the planes
    assert slabs[0].own0 == 0 and slabs[-1].own1 == shape[0]
    assert all(a.own1 == b.own0 for a, b in zip(slabs, slabs[1:]))


@pytest.mark.parametrize("mode", [3, 7])
def test_loopback_slabs_wave_forms(mode):
    """the slab protocol over the one-wave-per-tile operations (what the HIP library runs by default)"""
    import sim
    from medpy_amd.slab import LoopbackExchange, solve_slabs
    shape = (40, 24, 16)
    w, tr, ref = _problem("sphere", shape)
    sim.set_wave_mode(mode)
    try:
        slabs = [sim.SimSlab(shape, r, 3) for r in range(3)]
        for s in slabs:
            s.load(w, tr)
        st = solve_slabs(slabs, LoopbackExchange(slabs))
        assert st["converged"] == 1
        np.testing.assert_array_equal(np.concatenate([s.finish()[0] for s in slabs], axis=0), ref)
    finally:
        sim.set_wave_mode(0)


@pytest.mark.parametrize("conn,halo_max", [(6, 1), (6, 3), (26, 1), (26, 2)])
def test_full_border_messages_defer_and_drain(conn, halo_max):
    """A compacted border message holds MgcLattice::halo_max_rec records and travels in one transfer of known size; a border
    tile that finds it full keeps its labels / flow for the next exchange (mgc_halo_pack_tile, mgc26_halo_pack_tile), and the
    schedule neither takes "nothing woke up" for a fixpoint nor starts a global relabel while any are left.  With one to
    three record slots nearly every exchange overflows: the cut must not care."""
    import sim
    from medpy_amd import synthetic
    from medpy_amd.slab import LoopbackExchange, solve_slabs
    from oracle import energy_numpy, pipeline
    shape = (40, 24, 24)
    if conn == 6:
        w, tr, ref = _problem("sphere", shape)
        slabs = [sim.SimSlab(shape, r, 3) for r in range(3)]
    else:
        s = synthetic.sphere(shape)
        wo = energy_numpy.boundary_weights_offsets(s["term"], s["image"], energy_numpy.forward_offsets(3, 26), s["sigma"])
        g = pipeline.build_graph(s["fg"], s["bg"], weights=wo, connectivity=26)
        tr = np.array([g.get_trcap(i) for i in range(s["fg"].size)])
        g.maxflow()
        ref = g.labels().reshape(shape).astype(bool)
        w = sim.weights26(shape, wo)
        slabs = [sim.SimSlab26(shape, r, 3) for r in range(3)]
    for s_ in slabs:
        s_.load(w, tr)
        s_.set_halo_max(halo_max)
    st = solve_slabs(slabs, LoopbackExchange(slabs))
    assert st["converged"] == 1
    np.testing.assert_array_equal(np.concatenate([s_.finish()[0] for s_ in slabs], axis=0), ref)
    print("halo_max_rec %d, %d-neighbourhood: %d exchanges, %d drains before a global relabel" % (halo_max, conn, st["exchanges"], st.get("deferred_drains", 0)))


def test_schedule_independent_of_slab_count():
    import sim
    from medpy_amd.slab import LoopbackExchange, solve_slabs
    shape = (48, 16, 24)
    w, tr, ref = _problem("sphere", shape)
    for nslabs in (1, 2, 3, 6):
        slabs = [sim.SimSlab(shape, r, nslabs) for r in range(nslabs)]
        for s in slabs:
            s.load(w, tr)
        st = solve_slabs(slabs, LoopbackExchange(slabs), rounds_per_relabel=3, max_cycles=2, max_sweeps=4)
        assert st["converged"] == 1
        np.testing.assert_array_equal(np.concatenate([s.finish()[0] for s in slabs], axis=0), ref)


WORKER = r'''
import os, sys
import numpy as np
ROOT = sys.argv[1]; out = sys.argv[2]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
import torch.distributed as dist
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
import sim
from medpy_amd import synthetic
from medpy_amd.slab import solve_slabs
from dist_exchange import DistExchange
from oracle import energy_numpy, pipeline
shape = (40, 24, 24)
conn = int(sys.argv[3]) if len(sys.argv) > 3 else 6
s = synthetic.sphere(shape)
if conn == 6:
    w = energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"])
    g = pipeline.build_graph(s["fg"], s["bg"], weights=w)
    slab = sim.SimSlab(shape, rank, world)
else:
    w = energy_numpy.boundary_weights_offsets(s["term"], s["image"], energy_numpy.forward_offsets(3, 26), s["sigma"])
    g = pipeline.build_graph(s["fg"], s["bg"], weights=w, connectivity=26)
    slab = sim.SimSlab26(shape, rank, world)
    w = sim.weights26(shape, w)
tr = np.array([g.get_trcap(i) for i in range(s["fg"].size)])
slab.load(w, tr)
st = solve_slabs([slab], DistExchange(slab))
lab, _ = slab.finish()
np.save(os.path.join(out, "labels_%d.npy" % rank), lab)
np.save(os.path.join(out, "range_%d.npy" % rank), np.array([slab.own0, slab.own1, st["converged"], st["exchanges"]]))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.timeout(300)
@pytest.mark.parametrize("conn", [6, 26])
def test_two_process_gloo_slabs(tmp_path, conn):
    """world_size 2, gloo, one slab per process: the N>1 path of bench.py / the multi-GPU run (both neighbourhoods)."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    port = 29500 + ((os.getpid() + conn) % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script), ROOT, str(tmp_path), str(conn)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    if conn == 6:
        _, _, ref = _problem("sphere", (40, 24, 24))
    else:
        from medpy_amd import synthetic
        from oracle import pipeline
        s = synthetic.sphere((40, 24, 24))
        ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term=s["term"], image=s["image"], sigma=s["sigma"], connectivity=26).labels
    parts, ranges = [], []
    for r in range(2):
        parts.append(np.load(tmp_path / ("labels_%d.npy" % r)))
        ranges.append(np.load(tmp_path / ("range_%d.npy" % r)))
    assert ranges[0][1] == ranges[1][0] and ranges[0][2] == 1 and ranges[1][2] == 1 and ranges[0][3] > 0
    np.testing.assert_array_equal(np.concatenate(parts, axis=0), ref)


@pytest.mark.parametrize("incremental", [True, False])
def test_incremental_relabel_across_slabs(incremental):
    """several global relabels (rounds_per_relabel=1): the incremental closure has to cross the slab borders (halo kind 2)
    and both schedules must end on the oracle's labels."""
    import sim
    from medpy_amd.slab import LoopbackExchange, solve_slabs
    shape = (48, 24, 24)
    for gen in ("sphere", "hard"):
        w, tr, ref = _problem(gen, shape)
        for nslabs in (2, 3):
            slabs = [sim.SimSlab(shape, r, nslabs) for r in range(nslabs)]
            for s in slabs:
                s.load(w, tr)
            st = solve_slabs(slabs, LoopbackExchange(slabs), rounds_per_relabel=1, max_cycles=1, max_sweeps=2,
                             incremental_relabel=incremental)
            assert st["converged"] == 1 and st["outer"] >= 3
            np.testing.assert_array_equal(np.concatenate([s.finish()[0] for s in slabs], axis=0), ref)


@pytest.mark.parametrize("gen,shape,nslabs", [("sphere", (32, 16, 24), 2), ("hard", (40, 24, 24), 3), ("sphere", (64, 16, 16), 8),
                                              ("sphere", (21, 17, 24), 2)])
def test_full_neighbourhood_slabs_match_oracle(gen, shape, nslabs):
    """26-neighbourhood across slabs (BASELINE config 5): pushes over the border accumulate in the ghost tiles and
    travel as halo kind 1 (mgc26_halo_pack_tile); labels must equal the BK oracle fed the 26-neighbour edge list."""
    import sim
    from medpy_amd import synthetic
    from medpy_amd.slab import LoopbackExchange, solve_slabs
    from oracle import energy_numpy, pipeline
    s = getattr(synthetic, gen)(shape)
    offs = energy_numpy.forward_offsets(3, 26)
    w = energy_numpy.boundary_weights_offsets(s["term"], s["image"], offs, s["sigma"])
    g = pipeline.build_graph(s["fg"], s["bg"], weights=w, connectivity=26)
    tr = np.array([g.get_trcap(i) for i in range(s["fg"].size)])
    g.maxflow()
    ref = g.labels().reshape(shape).astype(bool)
    w26 = sim.weights26(shape, w)
    slabs = [sim.SimSlab26(shape, r, nslabs) for r in range(nslabs)]
    for sl in slabs:
        sl.load(w26, tr)
    st = solve_slabs(slabs, LoopbackExchange(slabs), rounds_per_relabel=2)
    assert st["converged"] == 1
    np.testing.assert_array_equal(np.concatenate([sl.finish()[0] for sl in slabs], axis=0), ref)


@pytest.mark.parametrize("incremental", [True, False])
@pytest.mark.parametrize("gen,shape,nslabs", [("hard", (48, 24, 24), 3), ("sphere", (40, 24, 17), 2), ("ties", (32, 16, 16), 4)])
def test_full_neighbourhood_incremental_relabel_across_slabs(gen, shape, nslabs, incremental):
    """26-neighbourhood, many global relabels (one round each): the DIRTY / SUSPECT flags of the border tiles travel as halo
    kind 2 (mgc26_halo_pack_tile) and the closure over the 26 supporting neighbour tiles crosses the slab borders."""
    import sim
    from medpy_amd import synthetic
    from medpy_amd.slab import LoopbackExchange, solve_slabs
    from oracle import energy_numpy, pipeline
    s = getattr(synthetic, gen)(shape)
    offs = energy_numpy.forward_offsets(3, 26)
    w = energy_numpy.boundary_weights_offsets(s["term"], s["image"], offs, s["sigma"])
    g = pipeline.build_graph(s["fg"], s["bg"], weights=w, connectivity=26)
    tr = np.array([g.get_trcap(i) for i in range(s["fg"].size)])
    g.maxflow()
    ref = g.labels().reshape(shape).astype(bool)
    w26 = sim.weights26(shape, w)
    slabs = [sim.SimSlab26(shape, r, nslabs) for r in range(nslabs)]
    for sl in slabs:
        sl.load(w26, tr)
    st = solve_slabs(slabs, LoopbackExchange(slabs), rounds_per_relabel=1, max_sweeps=2, incremental_relabel=incremental)
    assert st["converged"] == 1 and st["outer"] >= 3
    np.testing.assert_array_equal(np.concatenate([sl.finish()[0] for sl in slabs], axis=0), ref)


@pytest.mark.parametrize("gen,shape,nslabs", [("sphere", (48, 24, 24), 3), ("hard", (40, 24, 17), 2)])
def test_full_neighbourhood_slabs_with_a_regional_term(gen, shape, nslabs):
    """Round 6: with a regional term the relabel passes mark tiles MGC_ST_SETTLED (every voxel at label 1 or 2) and neither visit nor
    wake them again, the first pass of a relabel reads no halo, and discharges keep a label only on watched supports -- across slab
    borders (ghost tiles are never settled; an owned border tile that settled returns at once when a border message wakes it):
    many global relabels, labels = the BK oracle's."""
    import sim
    from medpy_amd import synthetic
    from medpy_amd.slab import LoopbackExchange, solve_slabs
    from oracle import energy_numpy, pipeline
    s = getattr(synthetic, gen)(shape)
    r = synthetic.regional(shape)
    w = energy_numpy.boundary_weights_offsets(s["term"], s["image"], energy_numpy.forward_offsets(3, 26), s["sigma"])
    g = pipeline.build_graph(s["fg"], s["bg"], weights=w, connectivity=26, prob=r["prob"], alpha=r["alpha"])
    tr = np.array([g.get_trcap(i) for i in range(s["fg"].size)])
    g.maxflow()
    ref = g.labels().reshape(shape).astype(bool)
    w26 = sim.weights26(shape, w)
    slabs = [sim.SimSlab26(shape, k, nslabs) for k in range(nslabs)]
    for sl in slabs:
        sl.load(w26, tr)
    st = solve_slabs(slabs, LoopbackExchange(slabs), rounds_per_relabel=1, max_sweeps=1)
    assert st["converged"] == 1 and st["outer"] >= 3
    np.testing.assert_array_equal(np.concatenate([sl.finish()[0] for sl in slabs], axis=0), ref)


STORE_WORKER = r'''
import os, sys
import numpy as np
ROOT = sys.argv[1]; out = sys.argv[2]; rank = int(sys.argv[3]); world = int(sys.argv[4])
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
assert "torch" not in sys.modules
import sim
from medpy_amd import synthetic
from medpy_amd.rendezvous import FileStore
from medpy_amd.slab import StoreExchange, solve_slabs
from oracle import energy_numpy, pipeline
store = FileStore(rank, world, directory=os.path.join(out, "rdv"), timeout=120)
shape = (40, 24, 24)
s = synthetic.sphere(shape)
w = energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"])
g = pipeline.build_graph(s["fg"], s["bg"], weights=w)
tr = np.array([g.get_trcap(i) for i in range(s["fg"].size)])
slab = sim.SimSlab(shape, rank, world)
slab.load(w, tr)
got = store.broadcast(bytes(range(128)) if rank == 0 else b"", src=0, nbytes=128)
assert got == bytes(range(128))
assert store.allreduce([rank + 1.0, 2.0], "sum").tolist() == [world * (world + 1) / 2.0, 2.0 * world]
assert store.allreduce([float(rank)], "max").tolist() == [world - 1.0]
st = solve_slabs([slab], StoreExchange(slab, store))
lab, _ = slab.finish()
np.save(os.path.join(out, "labels_%d.npy" % rank), lab)
np.save(os.path.join(out, "range_%d.npy" % rank), np.array([slab.own0, slab.own1, st["converged"], st["exchanges"]]))
store.close()
assert "torch" not in sys.modules
'''


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_multi_process_slabs_over_the_file_store(tmp_path, world):
    """The out-of-band channel of the package (medpy_amd/rendezvous.py: a private directory of files, fixed framing, no sockets, no
    pickle, no PyTorch) and the development transport on top of it: broadcast of a 128-byte id as RcclExchange does it, sums and
    maxima of host scalars, a barrier, and a whole slab solve with the borders travelling through it -- one process per slab."""
    script = tmp_path / "worker.py"
    script.write_text(STORE_WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(tmp_path), str(r), str(world)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    for p in procs:
        out, err = p.communicate(timeout=280)
        assert p.returncode == 0, err[-3000:]
    _, _, ref = _problem("sphere", (40, 24, 24))
    parts = [np.load(tmp_path / ("labels_%d.npy" % r)) for r in range(world)]
    ranges = [np.load(tmp_path / ("range_%d.npy" % r)) for r in range(world)]
    assert all(rg[2] == 1 and rg[3] > 0 for rg in ranges)
    np.testing.assert_array_equal(np.concatenate(parts, axis=0), ref)
    assert not os.path.exists(tmp_path / "rdv")  # the store cleaned up after itself


def test_file_store_refuses_a_directory_others_can_write(tmp_path):
    from medpy_amd.rendezvous import FileStore
    d = tmp_path / "open"
    d.mkdir(mode=0o777)
    os.chmod(d, 0o777)
    with pytest.raises(RuntimeError):
        FileStore(0, 1, directory=str(d))
