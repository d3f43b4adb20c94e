"""Region (label) energy terms for ``graph_from_labels``: reference medpy/graphcut/energy_label.py:33-461, same names,
same ``(graph, label_image, args)`` plug-in signature, same exceptions.

When the graph is this package's ``GCGraph`` facade the terms *record* themselves and the region adjacency graph is
built in MI355X HBM (``msg_add_label_edges``: border pixel pairs -> compaction -> stable sort -> ordered sums; region
sums by sort + reduce-by-key).  A foreign graph object (anything with ``set_nweight`` / ``set_tweight``, like the
recording graph of reference tests/graphcut_/energy_label.py:196-215) is driven call by call in the reference's order.
"""
import math
import sys

import numpy

from .generate import _consecutive_region_ids

__all__ = ["boundary_difference_of_means", "boundary_stawiaski", "boundary_stawiaski_directed", "regional_atlas"]


def _axis_pairs(arr, axis):
    a = [slice(None)] * arr.ndim
    b = [slice(None)] * arr.ndim
    a[axis] = slice(None, -1)
    b[axis] = slice(1, None)
    return arr[tuple(a)], arr[tuple(b)]


def _prepare(label_image):
    label_image = numpy.asarray(label_image)
    if label_image.flags["F_CONTIGUOUS"]:
        label_image = numpy.ascontiguousarray(label_image)
    _consecutive_region_ids(label_image)  # 1..n without gaps, or AttributeError (energy_label.py:451-461)
    return label_image


def boundary_difference_of_means(graph, label_image, original_image):
    r"""Boundary term based on the difference of means between adjacent image regions (reference energy_label.py:33-120):
    :math:`w(x,y) = \max(1 - |\bar I_x - \bar I_y| / \alpha, \epsilon)` with :math:`\alpha = |\max\bar I - \min\bar I|`."""
    label_image = _prepare(label_image)
    original_image = numpy.asarray(original_image)
    if hasattr(graph, "record_label_boundary"):
        return graph.record_label_boundary("difference_of_means", label_image, original_image)
    # foreign graph: one set_nweight per adjacent region pair
    lab = label_image.ravel()
    sums = numpy.bincount(lab, weights=numpy.asarray(original_image, dtype=numpy.float64).ravel())
    means = sums[1:] / numpy.bincount(lab)[1:]  # scipy.ndimage.mean, energy_label.py:88
    max_difference = float(abs(min(means) - max(means)))
    seen = set()
    for dim in range(label_image.ndim):
        kf, kt = _axis_pairs(label_image, dim)
        valid = kf != kt
        for a, b in zip(numpy.minimum(kf, kt)[valid].tolist(), numpy.maximum(kf, kt)[valid].tolist()):
            if (a, b) in seen:
                continue
            seen.add((a, b))
            if 0.0 == max_difference:
                value = sys.float_info.min
            else:
                value = max(1.0 - abs(means[a - 1] - means[b - 1]) / max_difference, sys.float_info.min)
            graph.set_nweight(a - 1, b - 1, value, value)


def boundary_stawiaski(graph, label_image, gradient_image):
    r"""Boundary term of Stawiaski et al. (reference energy_label.py:123-214): every pixel pair across a region border
    adds :math:`(1 / (1 + \max(|g_p|, |g_q|)))^2` to the edge between the two regions."""
    label_image = _prepare(label_image)
    gradient_image = numpy.asarray(gradient_image)
    if hasattr(graph, "record_label_boundary"):
        return graph.record_label_boundary("stawiaski", label_image, gradient_image)
    for dim in range(label_image.ndim):
        kf, kt = _axis_pairs(label_image, dim)
        gf, gt = _axis_pairs(gradient_image, dim)
        valid = kf != kt
        gradient_max = numpy.maximum(numpy.abs(gf), numpy.abs(gt))[valid]
        for k1, k2, val in zip(numpy.minimum(kf, kt)[valid], numpy.maximum(kf, kt)[valid], gradient_max):
            weight = max(math.pow(1.0 / (1.0 + val), 2), sys.float_info.min)
            graph.set_nweight(k1 - 1, k2 - 1, weight, weight)


def boundary_stawiaski_directed(graph, label_image, xxx_todo_changeme):
    r"""Directed variant of the Stawiaski term (reference energy_label.py:217-353): the arc from the brighter to the
    darker voxel's region (``directedness > 0``) or from the darker to the brighter one (``< 0``) is strengthened by
    ``abs(directedness)`` (capped at 1).

    The reference raises ``TypeError`` for ``directedness >= 0`` (it calls its five-parameter helper with four
    arguments, :304, :347); here that branch does what its code states.  For ``< 0`` the result equals the
    reference's, including NumPy's extra evaluation of the first pair of every axis (``numpy.vectorize``)."""
    (gradient_image, directedness) = xxx_todo_changeme
    label_image = _prepare(label_image)
    gradient_image = numpy.asarray(gradient_image)
    if hasattr(graph, "record_label_boundary"):
        return graph.record_label_boundary("stawiaski_directed", label_image, gradient_image, directedness)
    beta = abs(directedness)
    for dim in range(label_image.ndim):
        k1s, k2s = _axis_pairs(label_image, dim)
        v1s, v2s = _axis_pairs(gradient_image, dim)
        def add(key1, key2, v1, v2):
            if key1 == key2:
                return
            weight = max(math.pow(1.0 / (1.0 + max(abs(v1), abs(v2))), 2), sys.float_info.min)
            strong = min(1, weight + beta)
            if (v1 > v2) == (directedness >= 0):
                graph.set_nweight(key1 - 1, key2 - 1, strong, weight)
            else:
                graph.set_nweight(key1 - 1, key2 - 1, weight, strong)

        if k1s.size:  # numpy.vectorize evaluates the first element once more, with NumPy scalars (energy_label.py:335-353)
            add(k1s.flat[0], k2s.flat[0], v1s.flat[0], v2s.flat[0])
        for key1, key2, v1, v2 in zip(k1s.ravel().tolist(), k2s.ravel().tolist(), v1s.ravel().tolist(), v2s.ravel().tolist()):
            add(key1, key2, v1, v2)


def regional_atlas(graph, label_image, xxx_todo_changeme1):
    r"""Regional term based on a probability atlas (reference energy_label.py:355-404): the sum of the atlas values
    under a region, times ``alpha``, becomes the region's source weight and, negated, its sink weight."""
    (probability_map, alpha) = xxx_todo_changeme1
    label_image = numpy.asarray(label_image)
    probability_map = numpy.asarray(probability_map)
    _consecutive_region_ids(label_image)  # 1..n without gaps, or AttributeError (energy_label.py:451-461)
    nregions = int(label_image.max())
    if hasattr(graph, "merge_tweights"):
        from .graph import region_sums
        sums, _ = region_sums(label_image, probability_map, nregions)
        if probability_map.dtype == numpy.float32:
            # numpy.sum gave a float32 scalar and python-float * float32 stays float32 (energy_label.py:394-400, NEP 50)
            src = (numpy.float32(alpha) * sums.astype(numpy.float32)).astype(numpy.float64)
            snk = -src
        else:
            src, snk = alpha * sums, -1.0 * alpha * sums
        return graph.merge_tweights(numpy.arange(nregions), src, snk)
    lab = label_image.ravel()
    for rid in range(1, nregions + 1):
        weight = numpy.sum(probability_map.ravel()[lab == rid])
        graph.set_tweight(rid - 1, alpha * weight, -1.0 * alpha * weight)
