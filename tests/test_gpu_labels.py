"""-m gpu: the region (label) graph cut and the other sparse-graph users on the MI355X.
graph_from_labels + energy_label terms -> RAG built in HBM (msg_add_label_edges) -> sparse-graph solver, against what the
REFERENCE produced for the same inputs (tests/golden/reference_labels.npz) and against the BK oracle."""
import os

import numpy as np
import pytest

from oracle import bk, energy_label_numpy as eln, energy_numpy, pipeline

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "reference_labels.npz"))
CASES = ["l2d_f32", "l2d_f64", "l3d_f32", "l3d_i16"]


def _terms(case):
    from medpy_amd.graphcut import energy_label as el
    g = lambda k: GOLD["%s/%s" % (case, k)]
    return {
        "stawiaski": dict(boundary_term=el.boundary_stawiaski, boundary_term_args=g("gradient")),
        "stawiaski_directed_neg": dict(boundary_term=el.boundary_stawiaski_directed, boundary_term_args=(g("gradient"), -0.5)),
        "difference_of_means": dict(boundary_term=el.boundary_difference_of_means, boundary_term_args=g("image")),
        "stawiaski_atlas": dict(boundary_term=el.boundary_stawiaski, boundary_term_args=g("gradient"),
                                regional_term=el.regional_atlas, regional_term_args=(g("prob"), 0.5)),
    }


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("term", ["stawiaski", "stawiaski_directed_neg", "difference_of_means", "stawiaski_atlas"])
def test_graph_from_labels_matches_reference(case, term):
    from medpy_amd import graphcut
    lab = GOLD[case + "/labels"]
    n = int(lab.max())
    g = graphcut.graph_from_labels(lab, GOLD[case + "/fg"], GOLD[case + "/bg"], **_terms(case)[term])
    key = "%s/%s" % (case, term)
    ref_edges = GOLD[key + "/edges"]
    pairs = np.argwhere(ref_edges != 0)
    pairs = pairs[:: max(1, len(pairs) // 60)]  # a sample of the arcs + a few absent ones (get_edge is a per-call read-back)
    got = np.array([g.get_edge(int(a), int(b)) for a, b in pairs])
    want = ref_edges[pairs[:, 0], pairs[:, 1]]
    if term == "difference_of_means":
        np.testing.assert_allclose(got, want, rtol=1e-9)  # region means: tree-ordered sums (reduce-by-key) vs bincount order
    else:
        np.testing.assert_array_equal(got, want)  # duplicate sums in insertion order: bit-exact
    assert g.get_edge(0, n - 1) == ref_edges[0, n - 1]
    trcap = np.array([g.get_trcap(i) for i in range(n)])
    if term.endswith("atlas"):
        np.testing.assert_allclose(trcap, GOLD[key + "/trcap"], rtol=1e-6)
    else:
        np.testing.assert_array_equal(trcap, GOLD[key + "/trcap"])
    flow = g.maxflow()
    assert flow == pytest.approx(float(GOLD[key + "/flow"]), rel=1e-6 if term.endswith("atlas") else 1e-9)
    seg = np.array([0 if g.what_segment(i) == g.termtype.SINK else 1 for i in range(n)], dtype=np.uint8)
    np.testing.assert_array_equal(seg, GOLD[key + "/segments"])
    np.testing.assert_array_equal(g.labels().astype(np.uint8), GOLD[key + "/segments"])
    assert g.get_node_num() == n and g.stats()["arcs"] == 2 * np.count_nonzero(np.triu(ref_edges + ref_edges.T))


def test_positive_directedness_against_the_restatement():
    """the reference raises TypeError for directedness >= 0 (energy_label.py:304,347); pinned by the NumPy restatement only"""
    from medpy_amd import graphcut
    from medpy_amd.graphcut import energy_label as el
    case = "l3d_f32"
    lab, grad = GOLD[case + "/labels"], GOLD[case + "/gradient"]
    g = graphcut.graph_from_labels(lab, GOLD[case + "/fg"], GOLD[case + "/bg"], boundary_term=el.boundary_stawiaski_directed,
                                   boundary_term_args=(grad, 0.25))
    flow = g.maxflow()
    oflow, oseg, _ = eln.graphcut_labels(lab, GOLD[case + "/fg"], GOLD[case + "/bg"], "stawiaski_directed", (grad, 0.25))
    np.testing.assert_array_equal(g.labels(), oseg)
    assert flow == pytest.approx(oflow, rel=1e-9)


def test_larger_label_image_and_wrapper():
    """a 3-D watershed-like partition with ~4000 regions; graphcut_stawiaski (wrapper.py:239-310) end to end"""
    from medpy_amd import graphcut, synthetic
    shape = (48, 64, 56)
    s = synthetic.sphere(shape)
    idx = np.indices(shape)
    coarse = tuple((idx[d] + (idx[(d + 1) % 3] // 5)) // 4 for d in range(3))
    flat = np.ravel_multi_index(coarse, [int(c.max()) + 1 for c in coarse])
    regions = flat.astype(np.int64) * 7 + 3  # arbitrary ids: the wrapper relabels
    grad = np.abs(np.gradient(s["image"].astype(np.float64))[0]).astype(np.float32)
    seg = graphcut.graphcut_stawiaski(regions, grad, s["fg"], s["bg"])
    assert seg.shape == shape and seg.dtype == np.bool_
    from medpy_amd.graphcut.wrapper import relabel
    lab = relabel(regions)
    oflow, oseg, _ = eln.graphcut_labels(lab, s["fg"], s["bg"], "stawiaski", grad)
    np.testing.assert_array_equal(seg, np.concatenate([[False], oseg])[lab])
    assert 0 < seg.mean() < 1


def test_raw_graphdouble_calls_random_graphs():
    """SparseGraph driven like maxflow.GraphDouble (add_node / sum_edge / add_tweights / maxflow / what_segment)"""
    from medpy_amd.graphcut import GraphDouble
    rng = np.random.default_rng(11)
    for trial in range(6):
        n = int(rng.integers(2, 400))
        m = int(rng.integers(1, 6 * n))
        i, j = rng.integers(0, n, m), rng.integers(0, n, m)
        keep = i != j
        i, j = i[keep], j[keep]
        cap, rev = rng.random(i.size) + 1e-3, rng.random(i.size) * (rng.random(i.size) < 0.7) + 1e-3
        src = np.where(rng.random(n) < 0.2, rng.random(n) * 3, 0.0)
        snk = np.where(rng.random(n) < 0.2, rng.random(n) * 3, 0.0)
        o = bk.BKGraph(n, max(16, i.size))
        o.sum_edges(i, j, cap, rev)
        o.add_tweights(None, src, snk)
        oflow = o.maxflow()
        g = GraphDouble(n, i.size)
        assert g.add_node(n) == 0
        for a, b, c, r in zip(i.tolist(), j.tolist(), cap.tolist(), rev.tolist()):
            g.sum_edge(a, b, c, r)
        for u in range(n):
            if src[u] or snk[u]:
                g.add_tweights(u, float(src[u]), float(snk[u]))
        flow = g.maxflow()
        assert flow == pytest.approx(oflow, rel=1e-9, abs=1e-12)
        np.testing.assert_array_equal(g.labels().astype(np.uint8), o.labels())
        assert g.what_segment(0) == (g.termtype.SOURCE if o.labels()[0] else g.termtype.SINK)


@pytest.mark.parametrize("term", ["difference_linear", "difference_exponential", "difference_division", "difference_power",
                                  "maximum_linear", "maximum_exponential", "maximum_division", "maximum_power"])
def test_four_dimensional_voxel_graph(term):
    """graph_from_voxels on a 4-D image (ndim*2 = 8 neighbours): n-links generated in HBM for any number of
    axes (msg_add_lattice_edges), solved by the sparse-graph solver; oracle = the same per-axis weights into BK"""
    from medpy_amd import graphcut
    from medpy_amd.graphcut import energy_voxel as ev
    shape = (6, 7, 5, 4)
    rng = np.random.default_rng(3)
    idx = np.indices(shape)
    r = np.sqrt(sum((idx[d] - (shape[d] - 1) / 2.0) ** 2 for d in range(4)))
    image = (60.0 * (r < 2.5) + rng.normal(0, 8, shape)).astype(np.float32)
    fg = r < 1.0
    bg = np.zeros(shape, bool)
    bg[0], bg[-1] = True, True
    spacing = (1.0, 2.0, 0.5, 1.5) if term.endswith("power") else False
    sigma = 15.0
    fn = getattr(ev, "boundary_" + term)
    args = (image, spacing) if term.endswith("linear") else (image, sigma, spacing)
    g = graphcut.graph_from_voxels(fg, bg, boundary_term=fn, boundary_term_args=args)
    flow = g.maxflow()
    ref = pipeline.graphcut_voxel(fg, bg, term=term, image=image, sigma=sigma, spacing=spacing)
    np.testing.assert_array_equal(g.labels().reshape(shape), ref.labels)
    assert flow == pytest.approx(ref.flow, rel=1e-9)
    w = energy_numpy.boundary_weights(term, image, sigma, spacing)
    got = (g.get_edge(0, 1), g.get_edge(0, int(np.prod(shape[1:]))))
    want = (w[3].ravel()[0], w[0].ravel()[0])
    if term.endswith("exponential") or term.endswith("power"):  # device exp / pow are not correctly rounded
        assert got == pytest.approx(want, rel=1e-12)
    else:
        assert got == want


def _solve_dimacs(text):
    """parse a DIMACS max-flow file and solve it with the BK oracle -> (flow, {dimacs id: 0 sink side / 1 source side})"""
    arcs, n = {}, 0
    for line in text.splitlines():
        if line.startswith("p max"):
            n = int(line.split()[2])
        elif line.startswith("a "):
            _, a, b, c = line.split()
            arcs[(int(a), int(b))] = arcs.get((int(a), int(b)), 0.0) + float(c)
    g = bk.BKGraph(n - 2, max(16, len(arcs)))
    src, snk = np.zeros(n - 2), np.zeros(n - 2)
    done = set()
    for (a, b), c in arcs.items():
        if a == 1:
            src[b - 3] += c
        elif b == 2:
            snk[a - 3] += c
        elif (b, a) not in done:
            g.sum_edges([a - 3], [b - 3], [c], [arcs.get((b, a), 0.0)])
            done.add((a, b))
    g.add_tweights(None, src, snk)
    return g.maxflow(), g.labels()


def test_dimacs_export_of_device_graphs_round_trips():
    """a graph built in HBM, written as DIMACS (write.py:29-76 layout), solved by an independent solver"""
    import io as _io
    from medpy_amd import graphcut, synthetic
    from medpy_amd.graphcut import energy_label as el
    s = synthetic.sphere((10, 12, 9))
    g = graphcut.graph_from_voxels(s["fg"], s["bg"], boundary_term=graphcut.energy_voxel.boundary_difference_exponential,
                                   boundary_term_args=(s["image"], s["sigma"], False))
    f = _io.StringIO()
    graphcut.graph_to_dimacs(g, f)
    flow = g.maxflow()
    oflow, olabels = _solve_dimacs(f.getvalue())
    np.testing.assert_array_equal(g.labels().ravel().astype(np.uint8), olabels)
    case = "l3d_f32"
    lab = GOLD[case + "/labels"]
    g = graphcut.graph_from_labels(lab, GOLD[case + "/fg"], GOLD[case + "/bg"], boundary_term=el.boundary_stawiaski_directed,
                                   boundary_term_args=(GOLD[case + "/gradient"], -0.5))
    f = _io.StringIO()
    graphcut.graph_to_dimacs(g, f)
    g.maxflow()
    oflow, olabels = _solve_dimacs(f.getvalue())
    np.testing.assert_array_equal(g.labels().astype(np.uint8), olabels)
    np.testing.assert_array_equal(olabels, GOLD[case + "/stawiaski_directed_neg/segments"])


# ---- maxflow.GraphDouble / GraphFloat / GraphInt as the reference binds them (wrapper.cpp:27-134) ----
def _graph_types():
    from medpy_amd import graphcut
    return [graphcut.GraphDouble, graphcut.GraphFloat, graphcut.GraphInt]


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_get_edge_kat_of_the_reference(kind):
    """reference lib/maxflow/src/get_edge_test.py:17-59 (FIRST, SECOND, THIRD: RANDOM), on all three graph types: the graph
    grows past its constructor's node count, get_edge reads both directions, absent arcs read 0"""
    import random
    G = _graph_types()[kind]
    g = G(2, 1)
    g.add_node(3)
    g.add_edge(0, 1, 2, 2)
    g.add_edge(0, 2, 4, 5)
    assert [g.get_edge(0, 1), g.get_edge(1, 0), g.get_edge(0, 2), g.get_edge(2, 0), g.get_edge(1, 2), g.get_edge(2, 1)] == [2, 2, 4, 5, 0, 0]
    assert g.get_node_num() == 3
    g = G(2, 1)
    g.add_node(2)
    g.add_edge(0, 1, 2, 3)
    assert g.get_edge(0, 1) == 2 and g.get_edge(1, 0) == 3
    rnd = random.Random(5)
    nodes = 40
    g = G(nodes, nodes * (nodes - 1))
    g.add_node(nodes)
    want = {}
    for fr in range(nodes):
        for to in range(fr + 1, nodes):
            want[(fr, to)] = (rnd.randint(1, 10), rnd.randint(1, 10))
            g.add_edge(fr, to, want[(fr, to)][0], want[(fr, to)][1])
    for (fr, to), (c, r) in want.items():
        assert g.get_edge(fr, to) == c and g.get_edge(to, fr) == r
    if kind == 2:
        assert isinstance(g.get_edge(0, 1), int)


@pytest.mark.skipif(not bk.available("ref"), reason="oracle/_ref not built")
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_parallel_arcs_like_the_compiled_reference(kind):
    """add_edge creates PARALLEL arcs (graph.h:428-454): get_edge reports the pair added last (the first the list walk meets,
    graph.h:500-509), sum_edge adds to that pair, and the flow sees the sum of all of them -- every read-back and the cut
    compared with the unmodified reference driven through the same calls"""
    G = _graph_types()[kind]
    rng = np.random.default_rng(3 + kind)
    n = 30
    o = bk.BKGraph(n, 16, "ref")
    g = G(n, 16)
    assert g.add_node(n) == 0
    calls = []
    for k in range(400):
        a, b = (int(v) for v in rng.choice(n, 2, replace=False))
        c, r = (float(v) for v in rng.integers(0, 9, 2)) if kind else (float(rng.random()), float(rng.random()))
        which = "add" if rng.random() < 0.5 else "sum"
        calls.append((which, a, b, c, r))
    for which, a, b, c, r in calls:
        cc, rr = (int(c), int(r)) if kind == 2 else (c, r)
        if which == "add":
            o.add_edges([a], [b], [c], [r]); g.add_edge(a, b, cc, rr)
        else:
            o.sum_edges([a], [b], [c], [r]); g.sum_edge(a, b, cc, rr)
    for a in range(n):
        for b in range(n):
            if a != b:
                assert g.get_edge(a, b) == o.get_edge(a, b), (a, b)
    src = np.where(rng.random(n) < 0.3, rng.integers(1, 20, n), 0).astype(float)
    snk = np.where(rng.random(n) < 0.3, rng.integers(1, 20, n), 0).astype(float)
    o.add_tweights(None, src, snk)
    for u in range(n):
        if src[u] or snk[u]:
            g.add_tweights(u, int(src[u]) if kind == 2 else float(src[u]), int(snk[u]) if kind == 2 else float(snk[u]))
    oflow, flow = o.maxflow(), g.maxflow()
    assert flow == pytest.approx(oflow, rel=1e-6 if kind == 1 else 1e-12)
    np.testing.assert_array_equal(g.labels().astype(np.uint8), o.labels())  # the minimal sink set is unique, ties or not


def test_reset_returns_an_empty_graph():
    """Graph::reset, graph.cpp:46-60: no nodes, no arcs, no t-links, flow 0 -- and the object is usable again"""
    from medpy_amd.graphcut import GraphDouble
    g = GraphDouble(4, 4)
    g.add_node(4)
    g.add_edge(0, 1, 3.0, 1.0)
    g.add_tweights(0, 5.0, 0.0)
    g.add_tweights(1, 0.0, 5.0)
    assert g.maxflow() == 3.0
    g.reset()
    assert g.get_node_num() == 0 and g.get_arc_num() == 0 and g.get_trcap(0) == 0.0
    assert g.add_node(3) == 0
    g.add_edge(0, 2, 2.0, 2.0)
    g.add_tweights(0, 7.0, 0.0)
    g.add_tweights(2, 0.0, 5.0)
    assert g.maxflow() == 2.0 and g.get_edge(0, 1) == 0.0  # the new arc is the bottleneck; the arc of the old graph is gone
    assert list(g.labels()[:3]) == [True, True, False]


def test_graph_float_and_int_compute_in_their_type():
    """instances.inc:12-15: GraphFloat keeps float32 capacities and running sums, GraphInt integers; non-integers are refused
    by GraphInt as the Boost.Python signature would"""
    from medpy_amd.graphcut import GraphFloat, GraphInt
    f = GraphFloat(3, 2)
    f.add_node(3)
    f.sum_edge(0, 1, 0.1, 0.1)
    assert f.get_edge(0, 1) == float(np.float32(0.1))
    f.sum_edge(0, 1, 0.2, 0.0)
    assert f.get_edge(0, 1) == float(np.float32(np.float32(0.1) + np.float32(0.2)))
    f.add_tweights(0, 1.0, 0.0)
    f.add_tweights(1, 0.0, 1.0)
    assert f.maxflow() == float(np.float32(np.float32(0.1) + np.float32(0.2)))
    g = GraphInt(3, 2)
    g.add_node(3)
    g.add_edge(0, 1, 3, 0)
    g.add_edge(1, 2, 2, 0)
    g.add_tweights(0, 9, 0)
    g.add_tweights(2, 0, 9)
    flow = g.maxflow()
    assert flow == 2 and isinstance(flow, int)
    with pytest.raises(TypeError):
        g.add_edge(0, 2, 1.5, 0)


@pytest.mark.parametrize("kind", ["float", "int"])
def test_edges_added_after_a_solve_are_part_of_the_next(kind):
    """Graph<int> / Graph<float> take add_edge / sum_edge at any time (graph.h:428-480), also between two maxflow() calls: the
    second solve sees the new arcs (the capacities went to the device once only until round 6).  Against GraphDouble driven
    through the same calls and against the compiled reference."""
    from medpy_amd.graphcut import GraphDouble, GraphFloat, GraphInt
    from oracle import bk
    rng = np.random.default_rng(5)
    n = 40
    G = GraphFloat if kind == "float" else GraphInt
    g, d = G(n, 4), GraphDouble(n, 4)
    g.add_node(n); d.add_node(n)
    # (float: eighths, so that the float32 running sums of sum_edge are exact and GraphDouble / the reference see the same capacities)
    val = (lambda: float(rng.integers(1, 33)) / 8.0) if kind == "float" else (lambda: int(rng.integers(1, 9)))
    calls = []

    def both(name, *a):
        calls.append((name, a))
        getattr(g, name)(*a); getattr(d, name)(*a)

    for u in range(n):
        both("add_tweights", u, val(), val())
    for _ in range(3):  # three solves, edges added in between with both calls
        for _e in range(60):
            i, j = (int(v) for v in rng.integers(0, n, 2))
            if i != j:
                both("add_edge" if rng.random() < 0.5 else "sum_edge", i, j, val(), val())
        fg, fd = g.maxflow(), d.maxflow()
        assert abs(fg - fd) <= 1e-9 * max(1.0, abs(fd)), (fg, fd)
        assert np.array_equal(g.labels(), d.labels())
        # ... and the reference, told everything so far
        o = bk.BKGraph(n, 4 * len(calls))
        for name, a in calls:
            if name == "add_tweights":
                o.add_tweights(np.array([a[0]]), np.array([float(a[1])]), np.array([float(a[2])]))
            else:
                o.sum_edges(np.array([a[0]]), np.array([a[1]]), np.array([float(a[2])]), np.array([float(a[3])]))
        fo = o.maxflow()
        assert abs(fg - fo) <= 1e-9 * max(1.0, abs(fo)), (fg, fo)
        assert np.array_equal(g.labels().astype(np.uint8), o.labels())
