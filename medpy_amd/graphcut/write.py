"""DIMACS max-flow export; reference medpy/graphcut/write.py:29-76 (same text layout: source = 1, sink = 2).

Accepts the dict-based ``Graph`` (1-based node ids, written exactly like the reference does) and the solver objects of
this package (``VoxelGraph``, ``SparseGraph``: 0-based node ids, capacities read back from HBM), so that a graph built
on the GPU can be cross-checked with any third-party DIMACS solver."""
import numpy


def graph_to_dimacs(g, f):
    """Persists the supplied graph in valid DIMACS format into the file-like object ``f``."""
    f.write("c Created by medpy\n")
    f.write("c Oskar Maier, oskar.maier@googlemail.com\n")
    f.write("c\n")
    f.write("c problem line\n")
    if hasattr(g, "get_nweights") and hasattr(g, "get_edges"):  # dict Graph: node ids start at 1
        tlinks = [(node + 2, w[0], w[1]) for node, w in list(g.get_tweights().items())]
        nlinks = [(e[0] + 2, e[1] + 2, w[0], w[1]) for e, w in list(g.get_nweights().items())]
        nodes, nedges = g.get_node_count(), len(g.get_edges())
    else:
        nodes, tlinks, nlinks = _device_graph(g)
        nedges = len(nlinks)
    f.write("p max {} {}\n".format(nodes + 2, nedges))
    f.write("c source descriptor\n")
    f.write("n 1 s\n")
    f.write("c sink descriptor\n")
    f.write("n 2 t\n")
    f.write("c terminal arcs (t-weights)\n")
    for node, ws, wt in tlinks:
        if not 0 == ws:
            f.write("a 1 {} {}\n".format(node, ws))
        if not 0 == wt:
            f.write("a {} 2 {}\n".format(node, wt))
    f.write("c inter-node arcs (n-weights)\n")
    for a, b, w, wr in nlinks:
        if not 0 == w:
            f.write("a {} {} {}\n".format(a, b, w))
        if not 0 == wr:  # reversed weights have to follow directly in the next line
            f.write("a {} {} {}\n".format(b, a, wr))
    f.write("c end-of-file")


def _device_graph(g):
    """(node count, t-links, n-links) of a VoxelGraph / SparseGraph; DIMACS id of node i is i + 3"""
    nodes = g.get_node_num()
    tr = numpy.asarray(g.tweights(), dtype=numpy.float64).ravel()
    tlinks = [(int(i) + 3, float(tr[i]) if tr[i] > 0 else 0.0, float(-tr[i]) if tr[i] < 0 else 0.0) for i in numpy.flatnonzero(tr)]
    nlinks = []
    if hasattr(g, "arcs"):
        tail, head, cap = g.arcs()
        fwd = tail < head
        back = {(int(t), int(h)): float(c) for t, h, c in zip(tail[~fwd], head[~fwd], cap[~fwd])}
        for t, h, c in zip(tail[fwd].tolist(), head[fwd].tolist(), cap[fwd].tolist()):
            nlinks.append((t + 3, h + 3, c, back.get((h, t), 0.0)))
    else:  # voxel lattice: symmetric capacities per axis, energy_voxel.py:650-664 numbering
        shape = g._shape
        ids = numpy.arange(nodes, dtype=numpy.int64).reshape(shape)
        for axis in range(len(shape)):
            w = g.nweights(axis)
            sl = [slice(None)] * len(shape)
            sl[axis] = slice(None, -1)
            a = ids[tuple(sl)].ravel()
            stride = int(numpy.prod(shape[axis + 1:]))
            for p, c in zip(a.tolist(), w.ravel().tolist()):
                nlinks.append((p + 3, p + stride + 3, c, c))
    return nodes, tlinks, nlinks
