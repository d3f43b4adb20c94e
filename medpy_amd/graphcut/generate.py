"""``graph_from_voxels``: reference medpy/graphcut/generate.py:33-174, same signature, same
plug-in protocol, same exceptions; the graph it returns lives in MI355X HBM.
"""
import inspect
import logging

import numpy

from .graph import GCGraph

logger = logging.getLogger("medpy_amd.graphcut")


def graph_from_voxels(
    fg_markers,
    bg_markers,
    regional_term=False,
    boundary_term=False,
    regional_term_args=False,
    boundary_term_args=False,
    connectivity=None,
):
    """Create a graph-cut ready graph to segment an nD image using the voxel neighbourhood.

    Drop-in for ``medpy.graphcut.graph_from_voxels`` (reference generate.py:33-174): every
    voxel is a node (C-order flat index of the marker shape), n-links join the 2*ndim
    neighbours, ``regional_term(graph, regional_term_args)`` then
    ``boundary_term(graph, boundary_term_args)`` are invoked as plug-ins, and the markers are
    wired to the terminals with weight ``GCGraph.MAX`` (t-links accumulate, graph.h:416-425).

    Returns the solver object (stand-in for ``maxflow.GraphDouble``): ``maxflow()``,
    ``what_segment(i)``, ``termtype``, and the bulk ``labels()``.

    Raises ``AttributeError`` when a term is not a callable of exactly two parameters
    (generate.py:135-146).

    ``connectivity`` (extension, not in the reference): ``None`` / ``2*ndim`` = the reference's
    neighbourhood; ``3**ndim - 1`` (8 in 2-D, 26 in 3-D) = full neighbourhood, the built-in boundary
    terms then apply the same g(.) to every neighbour offset (spacing: Euclidean offset length).
    """
    fg_mask = numpy.asarray(fg_markers, dtype=numpy.bool_)
    bg_mask = numpy.asarray(bg_markers, dtype=numpy.bool_)
    shape = fg_mask.shape

    # the two plug-ins, each called as term(graph, args); an absent one records nothing
    regional = _checked_plugin(regional_term, "regional_term", 2)
    boundary = _checked_plugin(boundary_term, "boundary_term", 2)

    # one node per voxel (C-order flat index); the edge count only mirrors the reference's estimate, the lattice is implicit
    n_edges = _lattice_edge_count(shape)
    if logger.isEnabledFor(10):  # DEBUG: counting the markers is a pass over the volume each
        logger.debug("graph_from_voxels: shape %s -> %d nodes, %d n-links; %d source / %d sink markers",
                     shape, fg_mask.size, n_edges, numpy.count_nonzero(fg_mask), numpy.count_nonzero(bg_mask))
    graph = GCGraph(fg_mask.size, n_edges, shape=shape, connectivity=connectivity)

    # order matters for the merged t-links (graph.h:416-425): regional term, boundary term, then the hard constraints
    regional(graph, regional_term_args)
    boundary(graph, boundary_term_args)
    # The reference wires the markers id by id (generate.py:169-172: ravel().nonzero() -> set_source_nodes / set_sink_nodes).  Every
    # marker voxel appears once in such a list, so the masks themselves say the same thing: they go to the library as they are
    # (one byte per voxel) instead of as id lists that are sorted, checked for repeats and scattered into a mask again
    # (0.1 s of NumPy at 512^3).  An empty mask wires nothing, as the reference's count_nonzero guard has it.
    graph.record_markers(fg_mask, bg_mask)
    return graph.get_graph()


def _checked_plugin(term, name, nparams):
    """A term is optional (any false value); when given it must be callable with exactly ``nparams`` positional parameters --
    the reference's protocol (generate.py:135-146, 280-291), which raises AttributeError otherwise."""
    if not term:
        return (lambda graph, args: None) if nparams == 2 else (lambda graph, label_image, args: None)
    if not callable(term) or len(inspect.getfullargspec(term).args) != nparams:
        raise AttributeError("%s must be a callable taking exactly %d parameters" % (name, nparams))
    return term


def _lattice_edge_count(shape):
    """Edges of the 2*ndim lattice: along every axis of extent s > 1 each of the prod(shape) / s lines holds s - 1 of them.
    (The reference arrives at the same number through floating point, generate.py:363-383; it only sizes its allocation.)"""
    extents = [int(s) for s in shape]
    total = 1
    for s in extents:
        total *= s
    return sum(total // s * (s - 1) for s in extents if s > 1)


def graph_from_labels(
    label_image,
    fg_markers,
    bg_markers,
    regional_term=False,
    boundary_term=False,
    regional_term_args=False,
    boundary_term_args=False,
):
    """Create a graph-cut ready graph to segment an nD image using the region neighbourhood.

    Drop-in for ``medpy.graphcut.graph_from_labels`` (reference generate.py:177-338): every region of the label image
    (labels 1..n) is a node, regions that touch under the ``ndim*2`` voxel neighbourhood are joined by arcs whose
    weights the ``boundary_term(graph, label_image, boundary_term_args)`` plug-in supplies (see
    :mod:`medpy_amd.graphcut.energy_label`), ``regional_term(graph, label_image, regional_term_args)`` supplies
    t-weights, and the regions under the markers are wired to the terminals with ``GCGraph.MAX``.  The region adjacency
    graph is assembled and solved in MI355X HBM; the returned object is the stand-in for ``maxflow.GraphDouble``.

    Raises ``AttributeError`` for a malformed label image or terms that do not take three parameters."""
    regions = numpy.asarray(label_image)
    fg_mask = numpy.asarray(fg_markers, dtype=numpy.bool_)
    bg_mask = numpy.asarray(bg_markers, dtype=numpy.bool_)
    region_ids = _consecutive_region_ids(regions)

    regional = _checked_plugin(regional_term, "regional_term", 3)
    boundary = _checked_plugin(boundary_term, "boundary_term", 3)

    # node r - 1 stands for region r; the edge count is the reference's guess (generate.py:296-300) and sizes nothing here
    graph = GCGraph(region_ids.size, 10 * region_ids.size)
    regional(graph, regions, regional_term_args)
    boundary(graph, regions, boundary_term_args)
    graph.set_source_nodes(numpy.unique(regions[fg_mask]) - 1)
    graph.set_sink_nodes(numpy.unique(regions[bg_mask]) - 1)
    return graph.get_graph()


def _consecutive_region_ids(label_image):
    """The region ids of a label image, which must be exactly 1..n (reference generate.py:352-360, energy_label.py:451-461:
    AttributeError otherwise)."""
    ids = numpy.unique(label_image)
    if ids.size == 0 or ids[0] != 1 or ids[-1] != ids.size:
        raise AttributeError("the label image must hold the region ids 1..n without gaps (found %d distinct ids between %s and %s)"
                             % (ids.size, ids[0] if ids.size else "-", ids[-1] if ids.size else "-"))
    return ids
