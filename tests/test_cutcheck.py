"""CPU tier: the principled parity relaxation itself (oracle/cutcheck.py).  The ambiguity set is computed from the
reference solver's residual graph; these tests pin what it accepts and -- more importantly -- what it refuses."""
import numpy as np
import pytest

from oracle import bk, cutcheck, energy_numpy, pipeline


@pytest.mark.parametrize("kind", ["port", "ref"])
def test_ambiguity_set_of_a_tie(kind):
    """source -> 0 -(1)- 1 -(1)- 2 -> sink, plus a node 3 hanging off 1 with capacity 5: the arcs 0-1 and 1-2 tie, so {0} and
    {0,1,3} are both minimum cuts; nodes 1 and 3 are ambiguous, 0 and 2 are not."""
    if not bk.available(kind):
        pytest.skip("oracle kind %s not built here" % kind)
    g = bk.BKGraph(4, 3, kind)
    g.sum_edges([0, 1, 1], [1, 2, 3], [1.0, 1.0, 5.0], [1.0, 1.0, 5.0])
    g.add_tweights([0, 2], [10.0, 0.0], [0.0, 10.0])
    assert g.maxflow() == 1.0
    fs, ts, amb = cutcheck.ambiguity(g)
    assert fs.tolist() == [True, False, False, False]
    assert ts.tolist() == [False, False, True, False]
    assert amb.tolist() == [False, True, False, True]
    # both minimum cuts cost exactly 1 (the ambiguous nodes 1 and 3 move together); splitting them does not
    e = ([0, 1, 1], [1, 2, 3], [1.0, 1.0, 5.0], [1.0, 1.0, 5.0], [10.0, 0.0, -10.0, 0.0])
    for side in ([1, 0, 0, 0], [1, 1, 0, 1]):
        assert cutcheck.exact_cut_value(np.array(side, bool), *e) == 1
    for side in ([1, 1, 0, 0], [1, 0, 0, 1]):
        assert cutcheck.exact_cut_value(np.array(side, bool), *e) == 6


def _ties_case(shape=(40, 40, 40)):
    from medpy_amd import synthetic
    s = synthetic.ties(shape)
    cut = pipeline.graphcut_voxel(s["fg"], s["bg"], term=s["term"], image=s["image"], sigma=s["sigma"])
    i, j, ww = cutcheck.lattice_edges(shape, energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"]))
    tr = np.where(s["fg"], 65535.0, 0.0) - np.where(s["bg"], 65535.0, 0.0)
    return s, cut, (i, j, ww, ww, tr)


def test_reference_labels_are_the_sink_reachable_set_and_exactly_minimal():
    s, cut, exact = _ties_case((24, 24, 24))
    fs, ts, amb = cutcheck.ambiguity(cut.graph, tol=0.0)  # tol 0: the reference's own notion of "residual"
    np.testing.assert_array_equal(ts.reshape(cut.labels.shape), ~cut.labels)  # what_segment == SINK <=> can reach the sink
    assert abs(float(cutcheck.exact_cut_value(cut.labels, *exact)) - cut.flow) <= 1e-12 * cut.flow


def test_check_refuses_a_flipped_unambiguous_voxel():
    s, cut, exact = _ties_case()
    fs, ts, amb = cutcheck.ambiguity(cut.graph)
    assert 0 < amb.sum() < 10  # a handful of genuinely ambiguous voxels, not a licence
    labels = cut.labels.copy()
    assert cutcheck.assert_labels_equivalent(labels, cut, 0, exact=exact) == 0
    # flipping an ambiguous voxel keeps a minimum cut of the same capacity: accepted within the bound, refused beyond it
    v = int(np.flatnonzero(amb)[0])
    labels.flat[v] = not labels.flat[v]
    try:
        n = cutcheck.assert_labels_equivalent(labels, cut, 1, exact=exact)
        assert n == 1
    except AssertionError as err:  # an ambiguous voxel on its own need not be a min cut (its tie partners matter): capacity check
        assert "cut capacities differ" in str(err)
    with pytest.raises(AssertionError, match="bound"):
        cutcheck.assert_labels_equivalent(labels, cut, 0, exact=exact)
    # flipping a voxel that every minimum cut agrees on is refused, whatever the bound
    labels = cut.labels.copy()
    u = int(np.flatnonzero(fs & ~s["fg"].ravel())[0])
    labels.flat[u] = not labels.flat[u]
    with pytest.raises(AssertionError, match="NOT ambiguous"):
        cutcheck.assert_labels_equivalent(labels, cut, 1000, exact=exact)


def test_sub_ulp_capacities_are_below_the_rounding_granularity():
    """the reference floors zero weights at DBL_MIN (energy_voxel.py:113): a residual of 2.2e-308 next to weights of order 1 is
    not a way to the sink any floating point solver can rely on -- it belongs to the ambiguity set, a residual of 1e-3 does not"""
    g = bk.BKGraph(3, 2)
    g.sum_edges([0, 1], [1, 2], [1.0, 2.2250738585072014e-308], [1.0, 2.2250738585072014e-308])
    g.add_tweights([0, 2], [0.5, 0.0], [0.0, 10.0])
    g.maxflow()
    fs, ts, amb = cutcheck.ambiguity(g)
    assert not ts[1] or amb[1] or fs[1]  # node 1 is not "certainly sink side" through a 1e-308 arc
    g = bk.BKGraph(3, 2)
    g.sum_edges([0, 1], [1, 2], [1.0, 1e-3], [1.0, 1e-3])
    g.add_tweights([0, 2], [1e-4, 0.0], [0.0, 10.0])
    g.maxflow()
    fs, ts, amb = cutcheck.ambiguity(g)
    assert ts[1] and ts[0]  # residual 9e-4 of 1e-3: a real way to the sink


def test_exact_adjudication_names_the_canonical_label_set():
    """oracle/exact_maxflow.py: on an exact tie between two minimum cuts the reference's definition (complement of the sink tree,
    graph.h:561-571) picks the LARGEST source side; an integer-scaled Dinic finds it without a rounding, and `adjudicate` says which of
    two label sets is that one.  A chain s -> 0 -> 1 -> 2 -> t with two arcs of the same capacity: cutting either is minimum."""
    from fractions import Fraction
    from oracle import exact_maxflow as em
    i, j = np.array([0, 1]), np.array([1, 2])
    cap, rev = np.array([0.1, 0.1]), np.array([0.1, 0.1])
    tr = np.array([5.0, 0.0, -5.0])
    t = em.canonical_sink_side(3, i, j, cap, rev, tr)
    assert t.tolist() == [False, False, True]  # node 1 cannot reach the sink once 1 -> 2 is saturated: the larger source side
    small, large = np.array([True, False, False]), np.array([True, True, False])
    v = Fraction(0.1)
    verdict = em.adjudicate(small, large, 3, i, j, cap, rev, tr, v, v)
    assert verdict["canonical"] == "reference" and verdict["hip_vs_reference"] == {"capacity": "equal", "source_sides": "second_contains_first"}
    assert em.adjudicate(large, large, 3, i, j, cap, rev, tr, v, v)["canonical"] == "both"
    # against the compiled reference on random graphs with dyadic capacities (every sum exact: BK itself is canonical there)
    from oracle import bk
    rng = np.random.default_rng(3)
    for _ in range(10):
        n = 30
        a, b = rng.integers(0, n, 150), rng.integers(0, n, 150)
        keep = a != b
        a, b = a[keep], b[keep]
        c, r = rng.integers(0, 16, a.size) / 4.0, rng.integers(0, 16, a.size) / 4.0
        trc = rng.integers(-8, 9, n) / 2.0
        o = bk.BKGraph(n, a.size)
        o.sum_edges(a, b, c, r)
        o.add_tweights(None, np.maximum(trc, 0), np.maximum(-trc, 0))
        o.maxflow()
        assert ((~em.canonical_sink_side(n, a, b, c, r, trc)).astype(np.uint8) == o.labels()).all()
