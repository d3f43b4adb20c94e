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
    label_image = numpy.asarray(label_image)
    fg_markers = numpy.asarray(fg_markers, dtype=numpy.bool_)
    bg_markers = numpy.asarray(bg_markers, dtype=numpy.bool_)
    __check_label_image(label_image)

    if not regional_term:
        regional_term = __regional_term_label
    if not boundary_term:
        boundary_term = __boundary_term_label
    if not hasattr(regional_term, "__call__") or not 3 == len(inspect.getfullargspec(regional_term)[0]):
        raise AttributeError("regional_term has to be a callable object which takes three parameters.")
    if not hasattr(boundary_term, "__call__") or not 3 == len(inspect.getfullargspec(boundary_term)[0]):
        raise AttributeError("boundary_term has to be a callable object which takes three parameters.")

    nodes = len(numpy.unique(label_image))
    edges = 10 * nodes  # the reference's guess (generate.py:296-300); sizes nothing here
    graph = GCGraph(nodes, edges)

    regional_term(graph, label_image, regional_term_args)
    boundary_term(graph, label_image, boundary_term_args)

    graph.set_source_nodes(numpy.unique(label_image[fg_markers] - 1))  # node ids start at 0
    graph.set_sink_nodes(numpy.unique(label_image[bg_markers] - 1))
    return graph.get_graph()


def __check_label_image(label_image):
    """labels have to be 1..n without gaps (reference generate.py:352-360 / energy_label.py:451-461)"""
    encountered_indices = numpy.unique(label_image)
    expected_indices = numpy.arange(1, label_image.max() + 1)
    if not encountered_indices.size == expected_indices.size or not (encountered_indices == expected_indices).all():
        raise AttributeError("The supplied label image does either not contain any regions or they are not labeled consecutively starting from 1.")


def __regional_term_label(graph, label_image, regional_term_args):
    """Fake regional_term function with the appropriate signature."""
    return {}


def __boundary_term_label(graph, label_image, boundary_term_args):
    """Fake boundary_term function with the appropriate signature."""
    return {}


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
