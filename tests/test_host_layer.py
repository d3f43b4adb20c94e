"""CPU tier: host logic, C-ABI surface, hostsim (the solver's single-source tile ops run on the host)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from medpy_amd import _lib, build
    build.build_library()
    header = open(os.path.join(ROOT, "include", "medpy_hip.h")).read()
    declared = set(re.findall(r"\b(m[gs][cg]_[a-z_]+)\s*\(", header))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert callable(ge.build) and callable(ge.smoke)


def test_fails_loudly_without_gpu():
    from medpy_amd import _lib, graphcut
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_lib.MedpyHipError) as ei:
        graphcut.graph_from_voxels(np.zeros((4, 4), bool), np.zeros((4, 4), bool))
    assert ei.value.code == _lib.ERR_NO_DEVICE


def test_facade_validation_matches_reference_contract():
    """reference tests/graphcut_/graph.py:28-84 (ValueError contract of GCGraph)."""
    from medpy_amd.graphcut import GCGraph, graph_from_voxels
    g = GCGraph(10, 20)
    assert g.get_node_count() == 10 and g.get_edge_count() == 20 and g.get_nodes() == list(range(10))
    for bad in ((-1, 1, 1, 1), (0, 10, 1, 1), (3, 3, 1, 1), (0, 1, 0, 1), (0, 1, 1, -1)):
        with pytest.raises(ValueError):
            g.set_nweight(*bad)
    g.set_nweight(0, 1, float("nan"), 1.0)  # NaN passes `<= 0` exactly like the reference (graph.py:436)
    with pytest.raises(ValueError):
        g.set_tweight(10, 1, 1)
    g.set_tweight(0, -3, 2)  # t-weights may be negative
    with pytest.raises(ValueError):
        g.set_source_nodes([10])
    with pytest.raises(ValueError):
        g.set_sink_nodes([-1])
    with pytest.raises(AttributeError):
        graph_from_voxels(np.zeros((2, 2)), np.zeros((2, 2)), regional_term=lambda a, b, c: None)
    with pytest.raises(AttributeError):
        graph_from_voxels(np.zeros((2, 2)), np.zeros((2, 2)), boundary_term=3)


def test_split_marker():
    from medpy_amd.graphcut import split_marker
    fg, bg = split_marker(np.array([[0, 1, 2], [2, 1, 5]]))
    assert fg.tolist() == [[False, True, False], [False, True, False]]
    assert bg.tolist() == [[False, False, True], [True, False, False]]


def _sim_case(gen, shape, **kw):
    import sim
    from medpy_amd import synthetic
    from oracle import energy_numpy, pipeline
    s = getattr(synthetic, gen)(shape)
    term = kw.pop("term", s["term"])
    w = energy_numpy.boundary_weights(term, s["image"], s["sigma"])
    g = pipeline.build_graph(s["fg"], s["bg"], weights=w)
    tr = np.array([g.get_trcap(i) for i in range(s["fg"].size)])
    g.maxflow()
    ref = g.labels().reshape(shape)
    lab, st = sim.solve(shape, w, tr, **kw)
    return lab, ref, st


@pytest.mark.parametrize("gen,shape", [("sphere", (16, 16, 16)), ("sphere", (40, 40, 40)), ("hard", (32, 32, 32)),
                                       ("sphere", (9, 21, 35)), ("sphere", (5, 8, 64))])
def test_hostsim_solver_matches_bk(gen, shape):
    """The solver's tile operations (same source as the HIP kernels) give the reference labels."""
    lab, ref, st = _sim_case(gen, shape)
    assert st["converged"] == 1
    np.testing.assert_array_equal(lab, ref)


@pytest.mark.parametrize("mode", [1, 2, 3, 7, 9])
@pytest.mark.parametrize("gen,shape", [("sphere", (16, 16, 16)), ("sphere", (40, 40, 40)), ("hard", (32, 32, 32)),
                                       ("sphere", (9, 21, 35)), ("sphere", (5, 8, 64)), ("ties", (24, 16, 16))])
def test_hostsim_wave_forms_match_bk(gen, shape, mode):
    """One-wave-per-tile discharge / relabel (mgc_wave_ops.inl, same source as k_discharge_w / k_relabel_w): every
    combination with the workgroup forms reaches the reference labels (ties: dyadic-free input, cut value only)."""
    kw = dict(wave_mode=mode)
    if gen == "ties":
        kw["term"] = "difference_linear"
    lab, ref, st = _sim_case(gen, shape, **kw)
    assert st["converged"] == 1
    if gen == "ties" and (lab != ref).any():
        # exact ties between minimum cuts: every differing voxel must be ambiguous in the oracle's residual graph and the
        # two cuts must cost exactly the same (oracle/cutcheck.py) -- never "a few voxels are fine"
        from medpy_amd import synthetic
        from oracle import cutcheck, energy_numpy, pipeline
        s = synthetic.ties(shape)
        cut = pipeline.graphcut_voxel(s["fg"], s["bg"], term="difference_linear", image=s["image"])
        np.testing.assert_array_equal(cut.labels, ref.astype(bool))
        i, j, ww = cutcheck.lattice_edges(shape, energy_numpy.boundary_weights("difference_linear", s["image"]))
        tr = np.where(s["fg"], 65535.0, 0.0) - np.where(s["bg"], 65535.0, 0.0)
        cutcheck.assert_labels_equivalent(lab.astype(bool), cut, exact=(i, j, ww, ww, tr))
    else:
        np.testing.assert_array_equal(lab, ref)


@pytest.mark.parametrize("repeat", [0, 3])
@pytest.mark.parametrize("gen,shape", [("sphere", (32, 32, 32)), ("hard", (24, 32, 40))])
def test_hostsim_wave_discharge_with_and_without_repeated_steps(gen, shape, repeat):
    """the two instances of the wave discharge (one in-plane push step per direction and sweep / repeated while somebody can push): never,
    and in every launch including the flood on radial labels -- the cut is the reference's either way"""
    import sim
    sim.set_repeat(repeat)
    try:
        lab, ref, st = _sim_case(gen, shape, wave_mode=3)
    finally:
        sim.set_repeat(1)
    assert st["converged"] == 1
    np.testing.assert_array_equal(lab, ref)


def test_hostsim_wave_schedule_independence():
    for kw in (dict(rounds=1, sweeps=1), dict(rounds=3, sweeps=4), dict(rounds=50, sweeps=64)):
        for mode in (3, 7):
            lab, ref, st = _sim_case("sphere", (24, 24, 24), wave_mode=mode, **kw)
            assert st["converged"] == 1
            np.testing.assert_array_equal(lab, ref)


def test_hostsim_schedule_independence():
    """Any schedule (rounds between global relabels, cycle / sweep budgets) reaches the same cut."""
    base = None
    for kw in (dict(), dict(rounds=1, cycles=1, sweeps=1), dict(rounds=3, cycles=2, sweeps=4), dict(rounds=50, cycles=16, sweeps=64)):
        lab, ref, st = _sim_case("sphere", (24, 24, 24), **kw)
        assert st["converged"] == 1
        np.testing.assert_array_equal(lab, ref)
        base = lab if base is None else base
        np.testing.assert_array_equal(lab, base)


def test_hostsim_dyadic_ties_exact():
    lab, ref, st = _sim_case("ties", (24, 24, 24), term="difference_division")  # sigma 1: weights 1, 1/2, 1/3, 1/4
    # 1/3 is not dyadic; use linear below for the exact case
    import sim
    from medpy_amd import synthetic
    from oracle import energy_numpy, pipeline
    s = synthetic.ties((24, 24, 24))
    img = s["image"].copy(); img.flat[0], img.flat[1] = 0.0, 4.0
    w = energy_numpy.boundary_weights("difference_linear", img)
    g = pipeline.build_graph(s["fg"], s["bg"], weights=w)
    tr = np.array([g.get_trcap(i) for i in range(img.size)])
    g.maxflow()
    lab, st = sim.solve(img.shape, w, tr)
    np.testing.assert_array_equal(lab, g.labels().reshape(img.shape))


@pytest.mark.parametrize("wave", [0, 16], ids=["workgroup_form", "wave_form"])
@pytest.mark.parametrize("gen,shape", [("sphere", (16, 16, 16)), ("hard", (24, 24, 24)), ("sphere", (9, 21, 30)), ("sphere", (1, 24, 40))])
def test_hostsim_full_neighbourhood_matches_bk(gen, shape, wave):
    """26-neighbourhood tile ops (mgc_tile_ops26.inl, same source as the k26_* kernels) vs the BK oracle fed the
    26-neighbour edge list."""
    import sim
    from medpy_amd import synthetic
    from oracle import energy_numpy, pipeline
    s = getattr(synthetic, gen)(shape)
    offs = energy_numpy.forward_offsets(3, 26)
    w = energy_numpy.boundary_weights_offsets(s["term"], s["image"], offs, s["sigma"])
    g = pipeline.build_graph(s["fg"], s["bg"], weights=w, connectivity=26)
    tr = np.array([g.get_trcap(i) for i in range(s["fg"].size)])
    g.maxflow()
    lab, st = sim.solve26(shape, w, tr, wave_mode=wave)
    assert st["converged"] == 1
    np.testing.assert_array_equal(lab, g.labels().reshape(shape))


@pytest.mark.parametrize("gen,shape", [("hard", (32, 32, 32)), ("ties", (24, 24, 24)), ("sphere", (24, 40, 17))])
@pytest.mark.parametrize("rounds,cycles,sweeps", [(1, -1, 2), (1, 1, 1), (2, -1, 4), (1, -1, -2), (2, -1, -4)])
def test_hostsim_full_neighbourhood_incremental_relabel(gen, shape, rounds, cycles, sweeps):
    """Short colour rounds force many global relabels; all but the first recompute only the SUSPECT tiles (26 supporting
    neighbour tiles in the status word, mgc26_suspect_tile).  Labels must still be the BK oracle's."""
    import sim
    from medpy_amd import synthetic
    from oracle import energy_numpy, pipeline
    s = getattr(synthetic, gen)(shape)
    offs = energy_numpy.forward_offsets(3, 26)
    w = energy_numpy.boundary_weights_offsets(s["term"], s["image"], offs, s["sigma"])
    g = pipeline.build_graph(s["fg"], s["bg"], weights=w, connectivity=26)
    tr = np.array([g.get_trcap(i) for i in range(s["fg"].size)])
    g.maxflow()
    wave = 16 if sweeps < 0 else 0  # (negative sweeps: the same budget through the one-wave-per-tile discharge, mgc_wave_ops26.inl)
    lab, st = sim.solve26(shape, w, tr, rounds=rounds, cycles=cycles, sweeps=abs(sweeps), wave_mode=wave)
    assert st["converged"] == 1 and st["outer"] >= 5
    np.testing.assert_array_equal(lab, g.labels().reshape(shape))


@pytest.mark.parametrize("conn,gen,shape,kw", [
    (6, "hard", (32, 32, 32), dict(rounds=1, cycles=-1, sweeps=2)),
    (6, "hard", (32, 32, 32), dict(rounds=1, cycles=1, sweeps=2)),
    (6, "ties", (24, 24, 24), dict(rounds=2, cycles=-1, sweeps=4)),
    (6, "sphere", (24, 40, 17), dict(rounds=1, cycles=-1, sweeps=2, wave_mode=3)),
    (6, "hard", (32, 32, 32), dict(rounds=1, cycles=-1, sweeps=2, wave_mode=7)),
    (26, "hard", (32, 32, 32), dict(rounds=1, cycles=-1, sweeps=2)),
    (26, "hard", (32, 32, 32), dict(rounds=1, cycles=1, sweeps=1)),
    (26, "ties", (24, 24, 24), dict(rounds=2, cycles=-1, sweeps=4)),
    (26, "sphere", (24, 40, 17), dict(rounds=1, cycles=-1, sweeps=2, wave_mode=16)),
    (26, "hard", (32, 32, 32), dict(rounds=1, cycles=-1, sweeps=2, wave_mode=16)),
])
def test_hostsim_incremental_relabels_leave_exact_distances(conn, gen, shape, kw):
    """A tile is recomputed by an incremental global relabel only if a label of it rose or one of its voxels lost its last
    residual arc one label down (or it stands on such a tile): after EVERY global relabel of a solve driven through many of them
    the labels must be the exact distances to the sink in the residual graph, in every form of the discharge."""
    import sim
    from medpy_amd import synthetic
    from oracle import energy_numpy
    s = getattr(synthetic, gen)(shape)
    tr = (np.where(s["fg"], 65535.0, 0.0) - np.where(s["bg"], 65535.0, 0.0)).ravel()
    sim.set_check_exact(1)
    sim.prof()
    try:
        if conn == 6:
            _, st = sim.solve(shape, energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"]), tr, **kw)
        else:
            w = energy_numpy.boundary_weights_offsets(s["term"], s["image"], energy_numpy.forward_offsets(3, 26), s["sigma"])
            _, st = sim.solve26(shape, w, tr, **kw)
        p = sim.prof()
    finally:
        sim.set_check_exact(0)
    assert st["converged"] == 1 and st["outer"] >= 5
    assert p[40] == st["outer"] and p[41] == 0


@pytest.mark.parametrize("gen,shape,kw", [
    ("sphere", (32, 32, 32), dict(rounds=1, cycles=-1, sweeps=1)),
    ("hard", (24, 40, 17), dict(rounds=1, cycles=-1, sweeps=2)),
    ("sphere", (32, 32, 32), dict(rounds=1, cycles=-1, sweeps=1, wave_mode=16)),
    ("ties", (24, 24, 24), dict(rounds=2, cycles=-1, sweeps=2, wave_mode=16)),
])
def test_hostsim_settled_tiles_of_a_regional_term_leave_exact_distances(gen, shape, kw):
    """Round 6, full neighbourhood: with a regional term every voxel holds a t-link, a relabel visit that finds every voxel of a tile at 1
    or 2 marks the tile MGC_ST_SETTLED (no later pass of that relabel looks at it, nobody wakes it; a tile that settles in the first
    pass never looks over its borders and carries no support bits), and a voxel that saturates an arc keeps its label only on a
    support its tile watches (mgc26_support_watched).  After EVERY global relabel the labels are the exact distances, and the cut is
    the reference's."""
    import sim
    from medpy_amd import synthetic
    from oracle import energy_numpy, pipeline
    s = getattr(synthetic, gen)(shape)
    r = synthetic.regional(shape)
    w = energy_numpy.boundary_weights_offsets(s["term"], s["image"], energy_numpy.forward_offsets(3, 26), s["sigma"])
    g = pipeline.build_graph(s["fg"], s["bg"], weights=w, connectivity=26, prob=r["prob"], alpha=r["alpha"])
    tr = np.array([g.get_trcap(i) for i in range(s["fg"].size)])
    g.maxflow()
    sim.set_check_exact(1)
    sim.prof()
    try:
        lab, st = sim.solve26(shape, w, tr, **kw)
        pr = sim.prof()
    finally:
        sim.set_check_exact(0)
    assert st["converged"] == 1 and st["outer"] >= 3
    assert pr[40] == st["outer"] and pr[41] == 0
    np.testing.assert_array_equal(lab, g.labels().reshape(shape))


@pytest.mark.parametrize("wave_mode,n", [(3, 112)])
def test_hostsim_floods_in_several_radial_cycles_leave_exact_distances(wave_mode, n, monkeypatch):
    """a flood cut into several cycles on radial labels (radial_rounds0 = 1, a relabel in between, labels lowered again) starts
    its later cycles from graphs with saturated arcs; a push that is admissible under the radial labels only can then open a residual arc that
    undercuts the exact label kept aside for the receiving voxel.  Tiles that take flow in on radial labels are DIRTY for that reason
    (mgcw_discharge_impl, mgc_discharge_tile): after EVERY global relabel of such a solve the labels are the exact distances.  The volume is large
    enough for the shortest source -> sink path to have some 45 hops."""
    import sim
    from medpy_amd import synthetic
    from oracle import energy_numpy
    shape = (n, n, n)
    s = synthetic.sphere(shape)
    tr = (np.where(s["fg"], 65535.0, 0.0) - np.where(s["bg"], 65535.0, 0.0)).ravel()
    monkeypatch.setenv("HOSTSIM_RADIAL", "1")
    monkeypatch.setenv("HOSTSIM_RADIAL_ROUNDS0", "1")
    monkeypatch.setenv("HOSTSIM_RADIAL_BUDGET", "16")
    sim.set_check_exact(1)
    sim.prof()
    try:
        _, st = sim.solve(shape, energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"]), tr, wave_mode=wave_mode)
        p = sim.prof()
    finally:
        sim.set_check_exact(0)
    assert st["converged"] == 1 and st["radial_cycles"] >= 3
    assert p[40] == st["outer"] and p[41] == 0


def test_dimacs_writer_text_equals_the_reference_layout():
    """reference medpy/graphcut/write.py:29-76 on the dict Graph (graph.py:31-264); expected text written out by hand from
    the reference's format strings"""
    import io as _io
    from medpy_amd.graphcut import Graph, graph_to_dimacs
    g = Graph()
    g.set_nodes(3)
    g.set_source_nodes([1])
    g.set_sink_nodes([3])
    g.set_nweights({(1, 2): (0.5, 0.25), (2, 3): (2, 0)})
    g.add_tweights({2: (0.125, 0)})
    assert g.inconsistent() is False and g.get_nodes() == [1, 2, 3] and g.get_edges() == [(1, 2), (2, 3)]
    f = _io.StringIO()
    graph_to_dimacs(g, f)
    assert f.getvalue() == ("c Created by medpy\nc Oskar Maier, oskar.maier@googlemail.com\nc\nc problem line\np max 5 2\n"
                            "c source descriptor\nn 1 s\nc sink descriptor\nn 2 t\nc terminal arcs (t-weights)\n"
                            "a 1 3 65535\na 5 2 65535\na 1 4 0.125\nc inter-node arcs (n-weights)\n"
                            "a 3 4 0.5\na 4 3 0.25\na 4 5 2\nc end-of-file")
    g.set_nweights({(1, 2): (1, 1), (2, 1): (1, 1), (1, 9): (1, 1)})
    assert g.inconsistent() == ["The reversed edges of (1, 2) is also in the n-weights.", "The reversed edges of (2, 1) is also in the n-weights.",
                                "Node 9 in edge (1, 9) but not in nodes."]  # the reference's wording (graph.py:241-264)


def test_merge_tweights_equals_call_by_call_add_tweights():
    """GCGraph.merge_tweights (vectorised, distinct ids) must leave exactly what the reference's per-node
    set_tweight -> Graph::add_tweights sequence leaves (graph.py:490-498, graph.h:416-425): tr_cap and the flow constant"""
    from medpy_amd.graphcut import GCGraph
    rng = np.random.default_rng(0)
    n = 300
    a, b = GCGraph(n, 10), GCGraph(n, 10)
    for rnd in range(3):  # three batches on top of each other: regional-like, then fg-like, then bg-like
        ids = rng.permutation(n)[: int(rng.integers(1, n))]
        src = rng.random(ids.size) * (rng.random(ids.size) < 0.7) * 10
        snk = rng.random(ids.size) * (rng.random(ids.size) < 0.7) * 10 - (rnd == 0) * 3  # negative sink weights occur (regional_atlas)
        a.merge_tweights(ids, src, snk)
        for i, s, t in zip(ids.tolist(), src.tolist(), snk.tolist()):
            b.set_tweight(i, s, t)
    ta, tb = a._GCGraph__tr, b._GCGraph__tr
    np.testing.assert_array_equal(ta, tb)
    assert a._GCGraph__flow_const == b._GCGraph__flow_const
    with pytest.raises(ValueError):
        a.merge_tweights([n], [1.0], [0.0])


def test_graph_from_labels_argument_checks():
    """reference generate.py:266-291 / energy_label.py:451-461: malformed label images and terms are rejected before
    anything touches the GPU"""
    from medpy_amd.graphcut import graph_from_labels
    good = np.asarray([[1, 1, 2], [3, 3, 2]])
    m = np.zeros(good.shape, bool)
    for bad in ([[1, 4, 8], [1, 3, 10]], [[2, 3, 4], [2, 3, 4]], [[0, 1], [1, 2]]):
        with pytest.raises(AttributeError):
            graph_from_labels(bad, m, m)
    with pytest.raises(AttributeError):
        graph_from_labels(good, m, m, boundary_term=lambda g, args: None)
    with pytest.raises(AttributeError):
        graph_from_labels(good, m, m, regional_term=lambda g, l, a, extra: None)


@pytest.mark.parametrize("gen,shape", [("sphere", (40, 40, 40)), ("sphere", (24, 40, 17)), ("hard", (32, 32, 32)), ("ties", (24, 16, 16))])
def test_radial_labels_are_a_valid_labelling(gen, shape):
    """mgc_dt_ops.inl, radial labels of the flood phase: d = min(exact, max(1, C - hops from the source)) must be a VALID
    push-relabel labelling -- never above the exact distance, at most 1 on sink-linked voxels, and d(u) <= d(v) + 1 along every
    lattice arc (all arcs are residual when the transform runs) -- and below the exact labels somewhere, or it does nothing."""
    import sim
    from medpy_amd import synthetic
    from oracle import energy_numpy
    s = getattr(synthetic, gen)(shape)
    w = energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"])
    tr = (np.where(s["fg"], 65535.0, 0.0) - np.where(s["bg"], 65535.0, 0.0)).ravel()
    _, h_exact, _ = sim.first_relabel(shape, w, tr, 1)
    _, h_rad, _ = sim.first_relabel(shape, w, tr, 7)
    g = [(d + 7) // 8 for d in shape]

    def untile(h):  # [tiles, 512] -> padded volume
        v = h.reshape(g[0], g[1], g[2], 8, 8, 8).transpose(0, 3, 1, 4, 2, 5).reshape(g[0] * 8, g[1] * 8, g[2] * 8)
        return v[:shape[0], :shape[1], :shape[2]].astype(np.int64)
    ex, rd = untile(h_exact), untile(h_rad)
    assert (rd <= ex).all() and (rd >= 1).all()
    assert (rd[s["bg"]] == 1).all()
    for ax in range(3):
        assert np.abs(np.diff(rd, axis=ax)).max() <= 1
    if gen == "sphere":
        assert (rd < ex).mean() > 0.5  # everywhere off the shortest source -> sink paths
        c = np.unravel_index(np.argmax(rd), rd.shape)  # the highest label sits on the source
        assert s["fg"][c]


def test_relabel_first_appearance_order():
    """reference medpy/filter/label.py:76-105: consecutive ids in order of first appearance (C order)"""
    from medpy_amd.graphcut.wrapper import relabel
    lab = np.asarray([[7, 7, 3], [9, 3, 7], [2, 2, 9]])
    np.testing.assert_array_equal(relabel(lab), [[1, 1, 2], [3, 2, 1], [4, 4, 3]])
    np.testing.assert_array_equal(relabel(lab, start=5), np.asarray([[1, 1, 2], [3, 2, 1], [4, 4, 3]]) + 4)


@pytest.mark.parametrize("gen,shape", [("sphere", (16, 16, 16)), ("sphere", (40, 24, 33)), ("hard", (9, 21, 35)), ("sphere", (5, 8, 64)),
                                       ("ties", (24, 16, 16)), ("sphere", (1, 30, 30))])
def test_first_relabel_as_distance_transform(gen, shape):
    """mgc_dt_ops.inl (same source as k_dt_scan / k_dt_finish): when every n-link of the volume is residual, six axis scans
    leave exactly the labels, the label-support bits and the ALLINF flags the relaxation passes of the first global relabel
    converge to; and the solve that starts from them reaches the reference labels."""
    import sim
    from medpy_amd import synthetic
    from oracle import energy_numpy, pipeline
    s = getattr(synthetic, gen)(shape)
    w = energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"])
    g = pipeline.build_graph(s["fg"], s["bg"], weights=w)
    tr = np.array([g.get_trcap(i) for i in range(s["fg"].size)])
    ran, h_dt, st_dt = sim.first_relabel(shape, w, tr, True)
    ran0, h_rx, st_rx = sim.first_relabel(shape, w, tr, False)
    assert ran and not ran0
    np.testing.assert_array_equal(h_dt, h_rx)
    keep = ~np.uint32(4 | 8)  # (DIRTY / SUSPECT are not the relabel's to set)
    np.testing.assert_array_equal(st_dt & keep, st_rx & keep)
    lab, ref, st = _sim_case(gen, shape)
    assert st["converged"] == 1
    if gen != "ties":
        np.testing.assert_array_equal(lab, ref)


def test_distance_transform_is_refused_when_an_arc_is_missing():
    """a saturated (zero) n-link inside the volume: the residual graph is not the full lattice, the passes must run"""
    import sim
    shape = (16, 16, 16)
    w = [np.ones((15, 16, 16)), np.ones((16, 15, 16)), np.ones((16, 16, 15))]
    w[1][3, 4, 5] = 0.0
    tr = np.zeros(shape)
    tr[0] = -1.0
    tr[8, 8, 8] = 5.0
    ran, h, st = sim.first_relabel(shape, w, tr, True)
    assert not ran
    assert int(h.reshape(2, 2, 2, 8, 8, 8)[1, 1, 1, 0, 0, 0]) == 9  # voxel (8, 8, 8): eight steps to plane 0, one into the sink


@pytest.mark.parametrize("gen,shape", [("sphere", (40, 40, 40)), ("hard", (32, 32, 32)), ("sphere", (9, 21, 35))])
def test_activation_by_status_word_reaches_the_same_cut(gen, shape):
    """mgcw_activate_tile(exact=false): with many candidate tiles the activation after a global relabel trusts the tiles' status
    words (excess under a finite label at the last visit) instead of reading their voxels; a stale word costs one empty visit
    and is cleared by it.  Forced for every activation here: same labels, and the solve still terminates."""
    import sim
    sim.lib().hostsim_set_act_exact(-1)
    try:
        lab, ref, st = _sim_case(gen, shape, wave_mode=1)
    finally:
        sim.lib().hostsim_set_act_exact(4096)
    assert st["converged"] == 1
    np.testing.assert_array_equal(lab, ref)


@pytest.mark.parametrize("gen,shape", [("sphere", (40, 40, 40)), ("hard", (48, 32, 40)), ("sphere", (9, 21, 35)), ("ties", (24, 16, 16)),
                                       ("hard", (17, 8, 50))])
def test_incremental_relabel_over_bricks(gen, shape):
    """mgc_brick_ops.inl (same source as k_relabel_b): the passes of an incremental global relabel over bricks of 2 x 2 x 2 tiles
    relax to the same fixpoint as the tile passes -- so every global relabel leaves the same labels, the discharges between
    them do the same, and the solve ends with the same cut -- in about half the passes."""
    import sim
    out = {}
    for bricks in (0, 1):
        sim.lib().hostsim_set_bricks(bricks)
        try:
            kw = dict(wave_mode=1)
            if gen == "ties":
                kw["term"] = "difference_linear"
            lab, ref, st = _sim_case(gen, shape, **kw)
        finally:
            sim.lib().hostsim_set_bricks(0)
        assert st["converged"] == 1
        out[bricks] = (lab, st)
        if gen != "ties":
            np.testing.assert_array_equal(lab, ref)
    np.testing.assert_array_equal(out[0][0], out[1][0])
    assert out[0][1]["outer"] == out[1][1]["outer"] and out[0][1]["phases"] == out[1][1]["phases"] and out[0][1]["discharge_tiles"] == out[1][1]["discharge_tiles"]
    print(gen, shape, "relabel passes: tiles", out[0][1]["relabel_passes"], "bricks", out[1][1]["relabel_passes"])


def test_boundary_table_is_the_reference_term_on_whole_numbers():
    """medpy_amd.graphcut.graph.boundary_table (what mgc_set_boundary_lut uploads): entry d is bit for bit what the NumPy
    restatement of the reference's term gives for an intensity difference (or maximum) of d; None where no table applies."""
    from medpy_amd.graphcut.graph import boundary_table
    from oracle import energy_numpy
    rng = np.random.default_rng(2)
    img = rng.integers(0, 900, (9, 11)).astype(np.uint16)
    for term, sigma in (("difference_exponential", 12.5), ("difference_power", 1.3)):
        t = boundary_table(term, img, sigma)
        assert t is not None and t.size == int(img.max()) - int(img.min()) + 1
        w = energy_numpy.boundary_weights(term, img, sigma)[1]
        d = np.abs(img[:, :-1].astype(float) - img[:, 1:].astype(float)).astype(int)
        assert t[d].tobytes() == np.ascontiguousarray(w, dtype=np.float64).tobytes()
    for term, sigma in (("maximum_exponential", 300.0), ("maximum_power", 0.8)):
        shifted = img.astype(np.int16) - 400
        t = boundary_table(term, shifted, sigma)
        assert t is not None and t.size == int(np.abs(shifted.astype(int)).max()) + 1  # indexed by max(|I_p|, |I_q|)
    assert boundary_table("difference_division", img, 3.0) is None  # IEEE-basic terms need no table
    assert boundary_table("difference_exponential", img.astype(np.float32) + 0.5, 3.0) is None  # not whole numbers
    assert boundary_table("difference_exponential", (img.astype(np.int64) * 1000), 3.0) is None  # range beyond the table limit


def test_table_facts_of_images_that_take_no_table():
    """graph.py:image_table_facts -- what boundary_table and the slabs' sync_boundary_table decide by.  A float image that holds anything
    but whole numbers is turned away at its first block and its range is never computed (two passes over the volume for nothing:
    26 ms of a 43 ms upload at 512^3 until round 5); NaN / inf / non-numeric images take no table either."""
    from medpy_amd.graphcut.graph import boundary_table, image_table_facts

    class Counting(np.ndarray):  # an image that reports every reduction over it
        calls = []

        def min(self, *a, **k):
            Counting.calls.append("min")
            return np.ndarray.min(self.view(np.ndarray), *a, **k)

        def max(self, *a, **k):
            Counting.calls.append("max")
            return np.ndarray.max(self.view(np.ndarray), *a, **k)

    noise = np.random.default_rng(0).normal(0.0, 10.0, (40, 40, 40)).astype(np.float32)
    f = image_table_facts("difference_exponential", noise.view(Counting))
    assert f is not None and f[0] is False and Counting.calls == []
    assert boundary_table("difference_exponential", noise, 15.0) is None
    whole = np.rint(noise)
    f = image_table_facts("difference_exponential", whole)
    assert f[0] is True and f[1] == whole.min() and f[2] == whole.max()
    assert boundary_table("difference_exponential", whole, 15.0) is not None
    late = whole.copy()
    late[-1, -1, -1] += 0.25  # (the last voxel of the last block)
    assert image_table_facts("difference_power", late)[0] is False
    for bad in (np.nan, np.inf):
        broken = whole.copy()
        broken[3, 4, 5] = bad
        assert boundary_table("difference_exponential", broken, 15.0) is None
    assert image_table_facts("difference_linear", whole) is None  # IEEE-basic term: no table, no question
    assert image_table_facts("difference_exponential", np.zeros((0, 3), np.float32)) is None
    assert image_table_facts("maximum_exponential", np.arange(12, dtype=np.int16).reshape(3, 4)) == (True, 0.0, 11.0)


@pytest.mark.parametrize("wave", [0, 16], ids=["workgroup_form", "wave_form"])
def test_hostsim_full_neighbourhood_on_radial_labels_reaches_the_same_cut(wave, monkeypatch):
    """Radial labels for the 26-neighbourhood: min(exact, max(1, C - floor(L1 distance from the source / 3))) is 1-Lipschitz along all 26
    arcs (mgc_radial_steps, mgc_dt_ops.inl), put on top of a first relabel by passes.  Measured and left off by default
    (profiles/r6_rejected_radial26_hostsim.jsonl); forced on here, the cut is the reference's."""
    import sim
    from medpy_amd import synthetic
    from oracle import energy_numpy, pipeline
    shape = (40, 40, 40)
    s = synthetic.sphere(shape)
    offs = energy_numpy.forward_offsets(3, 26)
    w = energy_numpy.boundary_weights_offsets(s["term"], s["image"], offs, s["sigma"])
    g = pipeline.build_graph(s["fg"], s["bg"], weights=w, connectivity=26)
    tr = np.array([g.get_trcap(i) for i in range(s["fg"].size)])
    g.maxflow()
    monkeypatch.setenv("HOSTSIM_RADIAL", "1")
    labels, st = sim.solve26(shape, w, tr, wave_mode=wave)
    assert st["converged"] == 1 and st["radial_cycles"] >= 1
    np.testing.assert_array_equal(labels.astype(bool), g.labels().reshape(shape).astype(bool))
