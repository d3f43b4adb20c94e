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
    fg_markers = numpy.asarray(fg_markers)
    bg_markers = numpy.asarray(bg_markers)
    logger.debug("Assuming %d nodes and %d edges for image of shape %s", fg_markers.size,
                 __voxel_4conectedness(fg_markers.shape), fg_markers.shape)
    graph = GCGraph(fg_markers.size, __voxel_4conectedness(fg_markers.shape), shape=fg_markers.shape, connectivity=connectivity)

    logger.info("Performing attribute tests...")
    fg_markers = numpy.asarray(fg_markers, dtype=numpy.bool_)
    bg_markers = numpy.asarray(bg_markers, dtype=numpy.bool_)

    if not regional_term:
        regional_term = __regional_term_voxel
    if not boundary_term:
        boundary_term = __boundary_term_voxel

    if not hasattr(regional_term, "__call__") or not 2 == len(inspect.getfullargspec(regional_term)[0]):
        raise AttributeError("regional_term has to be a callable object which takes two parameter.")
    if not hasattr(boundary_term, "__call__") or not 2 == len(inspect.getfullargspec(boundary_term)[0]):
        raise AttributeError("boundary_term has to be a callable object which takes two parameters.")

    logger.debug("#nodes=%d, #hardwired-nodes source/sink=%d/%d", fg_markers.size,
                 numpy.count_nonzero(fg_markers), numpy.count_nonzero(bg_markers))

    logger.info("Computing and adding terminal edge weights...")
    regional_term(graph, regional_term_args)

    logger.info("Computing and adding inter-node edge weights...")
    boundary_term(graph, boundary_term_args)

    logger.info("Setting terminal weights for the markers...")
    if not 0 == numpy.count_nonzero(fg_markers):
        graph.set_source_nodes(fg_markers.ravel().nonzero()[0])
    if not 0 == numpy.count_nonzero(bg_markers):
        graph.set_sink_nodes(bg_markers.ravel().nonzero()[0])

    return graph.get_graph()


def __regional_term_voxel(graph, regional_term_args):
    """Fake regional_term function with the appropriate signature (generate.py:341-343)."""
    return {}


def __boundary_term_voxel(graph, boundary_term_args):
    """Fake boundary_term function with the appropriate signature (generate.py:351-354)."""
    return {}


def __voxel_4conectedness(shape):
    """Number of edges for the 2*ndim neighbourhood (generate.py:363-383)."""
    shape = list(shape)
    while 1 in shape:
        shape.remove(1)
    return int(round(sum([(dim - 1) / float(dim) for dim in shape]) * numpy.prod(shape)))
