"""End-to-end CPU checker for the voxel graph-cut path.  TEST INFRASTRUCTURE ONLY.

Restates reference medpy/graphcut/generate.py:graph_from_voxels (lines 33-174) + the CLI
read-out (bin/medpy_graphcut_voxel.py:172-182) on top of the bulk BK checkers of
``oracle/bk.py``: same call order (regional t-links, boundary n-links axis by axis, fg
markers, bg markers), same accumulation rules, whole arrays per call.
"""
import numpy

from . import bk, energy_numpy

MAX = 65535  # GCGraph.MAX, reference graph.py:288-291


def voxel_edge_count(shape):
    """__voxel_4conectedness, generate.py:363-383."""
    shape = [s for s in shape if s != 1]
    return int(round(sum((d - 1) / float(d) for d in shape) * numpy.prod(shape))) if shape else 0


def split_marker(marker, fg_id=1, bg_id=2):
    """wrapper.py:39-69."""
    marker = numpy.asarray(marker)
    return (marker == fg_id), (marker == bg_id)


class Cut:
    def __init__(self, flow, labels, graph, build_s=None, solve_s=None):
        self.flow, self.labels, self.graph = flow, labels, graph
        self.build_s, self.solve_s = build_s, solve_s


def build_graph(fg, bg, term=None, image=None, sigma=None, spacing=False, prob=None, alpha=None, kind=None,
                weights=None, connectivity=None):
    """graph_from_voxels with built-in terms.  ``weights`` overrides the boundary weights
    (used to inject device-computed capacities into the CPU solver: the cross-inject test)."""
    fg = numpy.asarray(fg, dtype=numpy.bool_)  # generate.py:125-126
    bg = numpy.asarray(bg, dtype=numpy.bool_)
    shape = fg.shape
    n = fg.size
    conn = connectivity or 2 * fg.ndim
    if conn == 2 * fg.ndim:
        g = bk.BKGraph(n, voxel_edge_count(shape), kind)
    else:
        g = bk.BKGraph(n, n * (conn // 2), kind)
    if prob is not None:  # generate.py:159 -> energy_voxel.py:61-65 -> graph.py:551-552
        src, snk = energy_numpy.regional_probability_tweights(prob, alpha)
        g.add_tweights(None, src, snk)
    if conn == 2 * fg.ndim:
        if weights is None and term is not None:  # generate.py:164
            weights = energy_numpy.boundary_weights(term, image, sigma, spacing)
        if weights is not None:
            wshape = numpy.asarray(weights[0]).shape
            lat_shape = list(wshape)
            lat_shape[0] += 1
            g.sum_lattice(lat_shape, weights)
    else:
        offs = energy_numpy.forward_offsets(fg.ndim, conn)
        if weights is None:
            weights = energy_numpy.boundary_weights_offsets(term, image, offs, sigma, spacing)
        strides = [int(numpy.prod(shape[k + 1:])) for k in range(fg.ndim)]
        ids = numpy.arange(n, dtype=numpy.int64).reshape(shape)
        for o in offs:
            w = numpy.asarray(weights[tuple(o)])
            m = ~numpy.isnan(w)
            i = ids[m]
            j = i + sum(ok * s for ok, s in zip(o, strides))
            g.sum_edges(i, j, w[m])
    if numpy.count_nonzero(fg):  # generate.py:169-170 -> graph.py:341-344
        idx = fg.ravel().nonzero()[0]
        g.add_tweights(idx, numpy.full(idx.size, float(MAX)), numpy.zeros(idx.size))
    if numpy.count_nonzero(bg):  # generate.py:171-172 -> graph.py:377-380
        idx = bg.ravel().nonzero()[0]
        g.add_tweights(idx, numpy.zeros(idx.size), numpy.full(idx.size, float(MAX)))
    return g


def graphcut_voxel(fg, bg, **kw):
    """graph_from_voxels + maxflow + label read-out; labels: bool array, 0 where SINK."""
    import time
    t0 = time.perf_counter()
    g = build_graph(fg, bg, **kw)
    t1 = time.perf_counter()
    flow = g.maxflow()
    t2 = time.perf_counter()
    labels = g.labels().astype(numpy.bool_).reshape(numpy.asarray(fg).shape)
    return Cut(flow, labels, g, t1 - t0, t2 - t1)
