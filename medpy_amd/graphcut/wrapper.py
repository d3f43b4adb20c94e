"""Convenience wrappers; reference medpy/graphcut/wrapper.py (split_marker :39-69, graphcut_stawiaski :239-310)."""
import numpy


class ArgumentError(Exception):
    """reference medpy/core/exceptions.py:31-34"""
    pass


def split_marker(marker, fg_id=1, bg_id=2):
    """(foreground mask, background mask) of a marker image: the voxels equal to ``fg_id`` / ``bg_id``.
    Contract of the reference's helper (medpy/graphcut/wrapper.py:39-69); two comparisons."""
    marker = numpy.asarray(marker)
    return marker == fg_id, marker == bg_id


def relabel(label_image, start=1):
    """consecutive labels from ``start`` in order of first appearance; reference medpy/filter/label.py:76-105"""
    label_image = numpy.asarray(label_image)
    flat = label_image.ravel()
    uniq, first, inverse = numpy.unique(flat, return_index=True, return_inverse=True)
    rank = numpy.empty(uniq.size, dtype=numpy.int64)
    rank[numpy.argsort(first, kind="stable")] = numpy.arange(uniq.size)
    return (rank[inverse] + start).reshape(label_image.shape).astype(label_image.dtype if label_image.dtype.kind in "iu" else numpy.int64)


def graphcut_stawiaski(regions, gradient=False, foreground=False, background=False):
    """Region graph cut with Stawiaski's boundary term in one call: ``(regions, gradient, foreground, background)`` (or one
    4-tuple holding them, as the reference accepts) -> boolean segmentation of the region image's shape.
    Contract: reference medpy/graphcut/wrapper.py:239-310 (incl. ``ArgumentError`` for differing shapes)."""
    from .energy_label import boundary_stawiaski
    from .generate import graph_from_labels
    if all(arg is False for arg in (gradient, foreground, background)):  # the one-argument form
        regions, gradient, foreground, background = regions
    regions, gradient = numpy.asarray(regions), numpy.asarray(gradient)
    masks = [numpy.asarray(m, dtype=numpy.bool_) for m in (foreground, background)]
    if len({a.shape for a in (regions, gradient, *masks)}) != 1:
        raise ArgumentError("All supplied images must be of the same shape.")
    regions = relabel(regions)  # ids 1..n in order of first appearance: node k - 1 is region k
    graph = graph_from_labels(regions, masks[0], masks[1], boundary_term=boundary_stawiaski, boundary_term_args=gradient)
    graph.maxflow()
    in_foreground = numpy.concatenate(([False], graph.labels())).astype(numpy.bool_)  # indexed by region id
    return in_foreground[regions]
