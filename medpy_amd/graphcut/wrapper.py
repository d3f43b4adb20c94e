"""Convenience wrappers; reference medpy/graphcut/wrapper.py (split_marker :39-69, graphcut_stawiaski :239-310)."""
import numpy


class ArgumentError(Exception):
    """reference medpy/core/exceptions.py:31-34"""
    pass


def split_marker(marker, fg_id=1, bg_id=2):
    """Split a marker image into fg / bg binary masks; reference wrapper.py:39-69."""
    img_marker = numpy.asarray(marker)
    img_fgmarker = numpy.zeros(img_marker.shape, numpy.bool_)
    img_fgmarker[img_marker == fg_id] = True
    img_bgmarker = numpy.zeros(img_marker.shape, numpy.bool_)
    img_bgmarker[img_marker == bg_id] = True
    return img_fgmarker, img_bgmarker


def relabel(label_image, start=1):
    """consecutive labels from ``start`` in order of first appearance; reference medpy/filter/label.py:76-105"""
    label_image = numpy.asarray(label_image)
    flat = label_image.ravel()
    uniq, first, inverse = numpy.unique(flat, return_index=True, return_inverse=True)
    rank = numpy.empty(uniq.size, dtype=numpy.int64)
    rank[numpy.argsort(first, kind="stable")] = numpy.arange(uniq.size)
    return (rank[inverse] + start).reshape(label_image.shape).astype(label_image.dtype if label_image.dtype.kind in "iu" else numpy.int64)


def graphcut_stawiaski(regions, gradient=False, foreground=False, background=False):
    """Executes a Stawiaski label graph cut; reference wrapper.py:239-310 (a single 4-tuple argument is unpacked like
    there).  Returns the segmentation as boolean array of the region image's shape."""
    from .energy_label import boundary_stawiaski
    from .generate import graph_from_labels
    if gradient is False and foreground is False and background is False:
        regions, gradient, foreground, background = regions
    img_region = numpy.asarray(regions)
    img_gradient = numpy.asarray(gradient)
    img_fg = numpy.asarray(foreground, dtype=numpy.bool_)
    img_bg = numpy.asarray(background, dtype=numpy.bool_)
    if not (img_region.shape == img_gradient.shape == img_fg.shape == img_bg.shape):
        raise ArgumentError("All supplied images must be of the same shape.")
    img_region = relabel(img_region)
    gcgraph = graph_from_labels(img_region, img_fg, img_bg, boundary_term=boundary_stawiaski, boundary_term_args=(img_gradient))
    gcgraph.maxflow()
    mapping = numpy.concatenate([[False], gcgraph.labels()])  # region id -> True where what_segment != SINK
    return mapping[img_region].astype(numpy.bool_)
