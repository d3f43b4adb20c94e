"""oracle/bk_maxflow.c (C restatement) must be indistinguishable from the unmodified reference
solver compiled in place (oracle/_ref/libbkref.so): same flow bits, labels, residual t-links."""
import numpy as np
import pytest

from oracle import bk, pipeline

pytestmark = pytest.mark.skipif(not bk.available("ref"), reason="oracle/_ref not built (no /root/reference)")


def _both(n, i, j, cap, rev, src, snk):
    out = []
    for kind in ("ref", "port"):
        g = bk.BKGraph(n, max(len(i), 1), kind)
        if len(i):
            g.sum_edges(i, j, cap, rev)
        g.add_tweights(None, src, snk)
        f = g.maxflow()
        out.append((f, g.labels(), np.array([g.get_trcap(k) for k in range(n)]), g.get_arc_num()))
    return out


def test_random_graphs_bitwise():
    rng = np.random.default_rng(7)
    for trial in range(400):
        n = int(rng.integers(2, 60)); m = int(rng.integers(0, 300))
        i = rng.integers(0, n, m); j = rng.integers(0, n, m)
        keep = i != j; i, j = i[keep], j[keep]
        if trial % 3 == 0:  # tie-heavy integers, zero reverse capacities
            cap = rng.integers(1, 5, i.size).astype(float); rev = rng.integers(0, 5, i.size).astype(float)
        else:  # 25 decades of dynamic range
            cap = rng.random(i.size) * 10.0 ** rng.integers(-20, 5, i.size); rev = rng.random(i.size)
        src = rng.random(n) * (rng.random(n) < 0.3) * 65535 ** (trial % 2)
        snk = rng.random(n) * (rng.random(n) < 0.3)
        a, b = _both(n, i, j, cap, rev, src, snk)
        assert a[0] == b[0]
        np.testing.assert_array_equal(a[1], b[1]); np.testing.assert_array_equal(a[2], b[2])
        assert a[3] == b[3]


def test_sum_edge_and_realloc():
    """reference lib/maxflow/src/sum_edge_test.py:20-79: repeated pairs accumulate; exceeding the edge hint reallocs."""
    for kind in ("ref", "port"):
        g = bk.BKGraph(4, 1, kind)
        g.sum_edges([0, 0, 1, 2, 0], [1, 1, 2, 3, 2], [1., 2., 1., 1., 4.], [0.5, 0.5, 1., 1., 4.])
        assert g.get_edge(0, 1) == 3.0 and g.get_edge(1, 0) == 1.0 and g.get_edge(3, 0) == 0.0
        assert g.get_arc_num() == 8
        g2 = bk.BKGraph(40, 1, kind)
        i = np.arange(39); g2.sum_edges(i, i + 1, np.ones(39))
        g2.add_tweights([0], [5.0], [0.0]); g2.add_tweights([39], [0.0], [5.0])
        assert g2.maxflow() == 1.0


@pytest.mark.parametrize("gen,shape", [("sphere", (40, 40, 40)), ("hard", (32, 32, 32)), ("ties", (28, 28, 28)),
                                       ("sphere", (9, 33, 17))])
def test_lattices_bitwise(gen, shape):
    from medpy_amd import synthetic
    s = getattr(synthetic, gen)(shape)
    cuts = [pipeline.graphcut_voxel(s["fg"], s["bg"], kind=k, term=s["term"], image=s["image"], sigma=s["sigma"])
            for k in ("ref", "port")]
    assert cuts[0].flow == cuts[1].flow
    np.testing.assert_array_equal(cuts[0].labels, cuts[1].labels)
