"""-m gpu: empty / degenerate inputs the reference tolerates: no markers, one-sided markers, single
voxels, singleton axes, 1-D volumes, repeated solves.  Labels against the BK oracle."""
import numpy as np
import pytest

from oracle import pipeline

pytestmark = pytest.mark.gpu


def _cut(image, fg, bg, term="difference_exponential", sigma=5.0):
    from medpy_amd import graphcut
    args = (image, False) if term.endswith("linear") else (image, sigma, False)
    g = graphcut.graph_from_voxels(fg, bg, boundary_term=getattr(graphcut.energy_voxel, "boundary_" + term), boundary_term_args=args)
    flow = g.maxflow()
    ref = pipeline.graphcut_voxel(fg, bg, term=term, image=image, sigma=sigma)
    np.testing.assert_array_equal(g.labels(), ref.labels)
    assert flow == pytest.approx(ref.flow, rel=1e-9, abs=1e-300)
    return g, flow


@pytest.mark.parametrize("shape", [(1,), (7,), (1, 1), (1, 9), (9, 1), (3, 1, 5), (1, 1, 1), (1, 1, 17), (2, 2, 2), (8, 8, 8), (9, 8, 7)])
def test_small_and_singleton_shapes(shape):
    rng = np.random.default_rng(sum(shape))
    image = rng.normal(0, 10, shape).astype(np.float32)
    fg = np.zeros(shape, bool); bg = np.zeros(shape, bool)
    fg.flat[0] = True
    if fg.size > 1:
        bg.flat[-1] = True
    _cut(image, fg, bg)


def test_no_markers_at_all():
    """reference generate.py:169-172 skips empty marker sets; with no terminals every node is free -> label 1"""
    image = np.random.default_rng(0).normal(0, 10, (6, 7, 8)).astype(np.float32)
    z = np.zeros(image.shape, bool)
    g, flow = _cut(image, z, z)
    assert g.labels().all() and flow == 0.0


def test_only_foreground_or_only_background():
    image = np.random.default_rng(1).normal(0, 10, (10, 9, 12)).astype(np.float32)
    z = np.zeros(image.shape, bool)
    m = np.zeros(image.shape, bool); m[2:4, 3:5, 1:6] = True
    g, _ = _cut(image, m, z)
    assert g.labels().all()          # no sink: nobody can reach it
    g, _ = _cut(image, z, m)
    assert not g.labels().any()      # no source: everything connected to the sink markers is sink side


def test_everything_marked():
    image = np.random.default_rng(2).normal(0, 10, (5, 6, 7)).astype(np.float32)
    fg = np.ones(image.shape, bool); bg = np.zeros(image.shape, bool)
    _cut(image, fg, bg)
    _cut(image, bg, fg)
    _cut(image, fg, fg)  # every voxel both: all t-links cancel (graph.h:416-425)


def test_repeated_solve_and_rebuild_are_idempotent():
    from medpy_amd import graphcut, synthetic
    s = synthetic.sphere((24, 24, 24))
    g = graphcut.graph_from_voxels(s["fg"], s["bg"], boundary_term=graphcut.energy_voxel.boundary_difference_exponential,
                                   boundary_term_args=(s["image"], s["sigma"], False))
    f1 = g.maxflow(); l1 = g.labels().copy()
    f2 = g.maxflow()
    assert f1 == f2
    g._build()
    f3 = g.maxflow()
    assert f3 == f1 and np.array_equal(g.labels(), l1)
    assert g.what_segment(0) == g.termtype.SINK and int(g.what_segment(int(np.flatnonzero(s["fg"].ravel())[0]))) == 0


def test_constant_image_and_huge_dynamic_range():
    """all weights equal (exp(0) = 1) -> massive ties with exact arithmetic; and weights spanning DBL_MIN..1"""
    shape = (12, 12, 12)
    fg = np.zeros(shape, bool); fg[5:7, 5:7, 5:7] = True
    bg = np.zeros(shape, bool); bg[0] = True
    _cut(np.zeros(shape, np.float32), fg, bg)
    image = np.zeros(shape, np.float64); image[:, :, 6:] = 1e6  # exp(-1e12/..) underflows -> DBL_MIN clamp (energy_voxel.py:299)
    _cut(image, fg, bg, sigma=1.0)


def test_repeated_explicit_edges_sum_in_call_order():
    """Graph::sum_edge (graph.h:457-480) adds an edge given several times in call order; three or more contributions with
    non-associative values tell an ordered sum from any other (an atomicAdd per contribution is not ordered)."""
    from medpy_amd import graphcut
    from oracle import bk
    shape = (3, 4, 5)
    n = int(np.prod(shape))
    rng = np.random.default_rng(5)
    # 40 edges, each repeated 3..6 times with values spread over 16 orders of magnitude, in shuffled call order and both orientations
    base_i = rng.integers(0, n - 1, 40)
    base_j = base_i + 1
    keep = (base_j % shape[2]) != 0  # x-neighbours inside a row
    base_i, base_j = base_i[keep], base_j[keep]
    ei, ej, cw, cr = [], [], [], []
    for a, b in zip(base_i, base_j):
        for _ in range(int(rng.integers(3, 7))):
            flip = rng.random() < 0.5
            ei.append(int(b if flip else a)); ej.append(int(a if flip else b))
            cw.append(float(10.0 ** rng.uniform(-8, 8))); cr.append(float(10.0 ** rng.uniform(-8, 8)))
    order = rng.permutation(len(ei))
    ei, ej, cw, cr = (np.asarray(v)[order] for v in (ei, ej, cw, cr))
    g = graphcut.GCGraph(n, len(ei), shape=shape)
    for a, b, w, r in zip(ei, ej, cw, cr):
        g.set_nweight(int(a), int(b), float(w), float(r))
    g.set_tweight(0, 5.0, 0.0); g.set_tweight(n - 1, 0.0, 5.0)
    vg = g.get_graph()
    o = bk.BKGraph(n, len(ei))
    o.sum_edges(ei, ej, cw, cr)
    o.add_tweights([0, n - 1], [5.0, 0.0], [0.0, 5.0])
    for a, b in zip(base_i, base_j):
        assert vg.get_edge(int(a), int(b)) == o.get_edge(int(a), int(b))  # bit-identical, not approximately
        assert vg.get_edge(int(b), int(a)) == o.get_edge(int(b), int(a))
    assert vg.maxflow() == pytest.approx(o.maxflow(), rel=1e-12)
    np.testing.assert_array_equal(vg.labels().ravel().astype(np.uint8), o.labels())


def test_explicit_edge_batches_of_growing_size_and_rebuild():
    """the C ABI accepts a new batch after every build, of any size (the device buffers grow), and a rebuild applies the
    stored batch again"""
    from medpy_amd.graphcut.graph import VoxelGraph
    shape = (4, 4, 4)
    g = VoxelGraph(shape)
    tr = np.zeros(64); tr[0], tr[63] = 3.0, -3.0
    g._set_tweights_merged(tr, 0.0)
    g._add_edges([0], [1], [1.0], [1.0])
    g._build()
    assert g.get_edge(0, 1) == 1.0
    i = np.arange(0, 63, dtype=np.int64); i = i[(i + 1) % 4 != 0]
    g._add_edges(i, i + 1, np.full(i.size, 2.0), np.full(i.size, 0.5))  # 48 edges after a batch of one
    g._build()
    assert g.get_edge(0, 1) == 2.0 and g.get_edge(1, 0) == 0.5 and g.get_edge(61, 62) == 2.0
    g._build()  # rebuild without re-adding: same capacities
    assert g.get_edge(0, 1) == 2.0 and g.get_edge(62, 61) == 0.5
    from medpy_amd._lib import MedpyHipError
    with pytest.raises(MedpyHipError):
        g._add_edges([0], [5], [1.0], [1.0])  # not lattice neighbours: refused when added, naming the edge


def test_plugin_edge_between_distant_voxels_goes_to_the_sparse_solver():
    """the reference accepts arbitrary node pairs in set_nweight (graph.py:382-440): a plug-in term that links two voxels which
    are not lattice neighbours is solved on the sparse-graph solver instead of being refused at build time"""
    from medpy_amd import graphcut
    from oracle import bk
    shape = (3, 3)

    def term(graph, args):
        for a in range(8):
            if (a + 1) % 3:
                graph.set_nweight(a, a + 1, 1.0, 1.0)
        for a in range(6):
            graph.set_nweight(a, a + 3, 0.5, 0.5)
        graph.set_nweight(0, 8, 7.0, 7.0)  # a long-range link

    fg = np.zeros(shape, bool); fg[0, 0] = True
    bg = np.zeros(shape, bool); bg[1, 1] = True
    g = graphcut.graph_from_voxels(fg, bg, boundary_term=term, boundary_term_args=(None,))
    o = bk.BKGraph(9, 13)
    o.sum_edges([0, 1, 3, 4, 6, 7], [1, 2, 4, 5, 7, 8], [1.0] * 6, [1.0] * 6)
    o.sum_edges(list(range(6)), [a + 3 for a in range(6)], [0.5] * 6, [0.5] * 6)
    o.sum_edges([0], [8], [7.0], [7.0])
    o.add_tweights([0], [65535.0], [0.0]); o.add_tweights([4], [0.0], [65535.0])
    assert g.maxflow() == pytest.approx(o.maxflow(), rel=1e-12)
    res = np.array([0 if g.termtype.SINK == g.what_segment(i) else 1 for i in range(9)], dtype=np.uint8)
    np.testing.assert_array_equal(res, o.labels())
