"""CPU restatement of the reference's REGION energy terms and of graph_from_labels.  TEST INFRASTRUCTURE ONLY
(imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by medpy_amd).

Each function returns the calls the reference term makes on the graph, as arrays in the reference's order:
``(i, j, cap, rev, once)`` for ``set_nweight(i, j, cap, rev)`` (``once`` marks calls that repeat the same weight for a
region pair and that the reference issues only once), and per-region t-weights for the regional term.

Follows reference medpy/graphcut/energy_label.py:
  boundary_difference_of_means   :33-120
  boundary_stawiaski             :123-214
  boundary_stawiaski_directed    :217-353   (incl. NumPy's extra evaluation of the first element, see below)
  regional_atlas                 :355-404
and medpy/graphcut/generate.py:177-338 (graph_from_labels).
"""
import sys

import numpy

DBL_MIN = sys.float_info.min


def _axis_pairs(arr, axis):
    a = [slice(None)] * arr.ndim
    b = [slice(None)] * arr.ndim
    a[axis] = slice(None, -1)
    b[axis] = slice(1, None)
    return arr[tuple(a)], arr[tuple(b)]


def _as_python_like(x):
    """float64 view of values that the reference handles as Python scalars"""
    return numpy.asarray(x, dtype=numpy.float64)


def stawiaski_edges(label_image, gradient_image):
    """energy_label.py:186-214: per border pixel pair, weight = max(pow(1/(1+max|g|), 2), DBL_MIN) added to (kmin, kmax).
    `val` is a NumPy scalar of the gradient's dtype, so for float32 images 1.0 / (1.0 + val) is evaluated in float32
    (NEP 50 scalar promotion of the installed NumPy 2.x); math.pow then works on the float64 value of that result."""
    label_image = numpy.ascontiguousarray(label_image)
    gradient_image = numpy.asarray(gradient_image)
    i, j, w = [], [], []
    for dim in range(label_image.ndim):
        kf, kt = _axis_pairs(label_image, dim)
        gf, gt = _axis_pairs(gradient_image, dim)
        valid = kf != kt
        gmax = numpy.maximum(numpy.abs(gf), numpy.abs(gt))[valid]
        if gmax.dtype == numpy.float32:
            y = (numpy.float32(1.0) / (numpy.float32(1.0) + gmax)).astype(numpy.float64)
        else:
            y = 1.0 / (1.0 + gmax.astype(numpy.float64))
        weight = numpy.maximum(y * y, DBL_MIN)
        i.append(numpy.minimum(kf, kt)[valid].astype(numpy.int64) - 1)
        j.append(numpy.maximum(kf, kt)[valid].astype(numpy.int64) - 1)
        w.append(weight)
    i, j, w = numpy.concatenate(i), numpy.concatenate(j), numpy.concatenate(w)
    return i, j, w, w.copy(), numpy.zeros(i.size, numpy.uint8)


def stawiaski_directed_edges(label_image, gradient_image, directedness):
    """energy_label.py:304-353.  numpy.vectorize is used without otypes, so NumPy evaluates the Python function once
    more on the first element of every axis to learn the output type (numpy/lib/_function_base_impl.py,
    vectorize._get_ufunc_and_otypes) -- with NumPy scalars of the image dtype, before the regular loop hands Python
    scalars over.  If that first pair crosses a region border its weight is therefore added twice."""
    label_image = numpy.ascontiguousarray(label_image)
    gradient_image = numpy.asarray(gradient_image)
    beta = abs(directedness)
    i, j, cap, rev = [], [], [], []
    for dim in range(label_image.ndim):
        k1, k2 = _axis_pairs(label_image, dim)
        v1, v2 = _axis_pairs(gradient_image, dim)
        k1, k2, v1n, v2n = k1.ravel(), k2.ravel(), v1.ravel(), v2.ravel()
        v1f, v2f = _as_python_like(v1n), _as_python_like(v2n)
        valid = k1 != k2
        y = 1.0 / (1.0 + numpy.maximum(numpy.abs(v1f), numpy.abs(v2f)))
        weight = numpy.maximum(y * y, DBL_MIN)
        wb = numpy.minimum(1.0, weight + beta)
        heavier_first = (v1f > v2f) if directedness >= 0 else ~(v1f > v2f)
        c = numpy.where(heavier_first, wb, weight)
        r = numpy.where(heavier_first, weight, wb)
        ii, jj, cc, rr = (k1[valid].astype(numpy.int64) - 1, k2[valid].astype(numpy.int64) - 1, c[valid], r[valid])
        if k1.size and valid[0]:  # the extra evaluation of element 0, NumPy scalars of the image dtype
            if v1n.dtype == numpy.float32:
                m = numpy.maximum(numpy.abs(v1n[0]), numpy.abs(v2n[0]))
                y0 = float(numpy.float32(1.0) / (numpy.float32(1.0) + m))
            else:
                y0 = 1.0 / (1.0 + max(abs(float(v1n[0])), abs(float(v2n[0]))))
            w0 = max(y0 * y0, DBL_MIN)
            wb0 = min(1.0, w0 + beta)
            c0, r0 = (wb0, w0) if heavier_first[0] else (w0, wb0)
            ii, jj = numpy.concatenate([ii[:1], ii]), numpy.concatenate([jj[:1], jj])
            cc, rr = numpy.concatenate([[c0], cc]), numpy.concatenate([[r0], rr])
        i.append(ii); j.append(jj); cap.append(cc); rev.append(rr)
    i, j = numpy.concatenate(i), numpy.concatenate(j)
    return i, j, numpy.concatenate(cap), numpy.concatenate(rev), numpy.zeros(i.size, numpy.uint8)


def region_means(label_image, image):
    """scipy.ndimage.mean(image, labels, index=unique labels): float64 bincount sums / counts (energy_label.py:88)"""
    lab = numpy.asarray(label_image).ravel()
    sums = numpy.bincount(lab, weights=numpy.asarray(image, dtype=numpy.float64).ravel())
    cnt = numpy.bincount(lab)
    return sums[1:] / cnt[1:]


def difference_of_means_edges(label_image, original_image):
    """energy_label.py:80-120: ONE weight max(1 - |mean_a - mean_b| / max_difference, DBL_MIN) per adjacent region pair.
    Returned per border pixel pair with once = 1 (the reference de-duplicates with a set, __compute_edges_nd :421-448)."""
    label_image = numpy.ascontiguousarray(label_image)
    means = region_means(label_image, original_image)
    max_difference = float(abs(means.min() - means.max()))
    i, j = [], []
    for dim in range(label_image.ndim):
        kf, kt = _axis_pairs(label_image, dim)
        valid = kf != kt
        i.append(numpy.minimum(kf, kt)[valid].astype(numpy.int64) - 1)
        j.append(numpy.maximum(kf, kt)[valid].astype(numpy.int64) - 1)
    i, j = numpy.concatenate(i), numpy.concatenate(j)
    if max_difference == 0.0:
        w = numpy.full(i.size, DBL_MIN)
    else:
        w = numpy.maximum(1.0 - numpy.abs(means[i] - means[j]) / max_difference, DBL_MIN)
    return i, j, w, w.copy(), numpy.ones(i.size, numpy.uint8)


def atlas_tweights(label_image, probability_map, alpha):
    """energy_label.py:389-400: weight = numpy.sum(map over the region) (accumulated in the map's dtype);
    set_tweight(rid - 1, alpha * weight, -1.0 * alpha * weight)"""
    import scipy.ndimage
    label_image = numpy.asarray(label_image)
    probability_map = numpy.asarray(probability_map)
    objects = scipy.ndimage.find_objects(label_image)
    src, snk = [], []
    for rid in range(1, len(objects) + 1):
        weight = numpy.sum(probability_map[objects[rid - 1]][label_image[objects[rid - 1]] == rid])
        src.append(float(alpha * weight))
        snk.append(float(-1.0 * alpha * weight))
    return numpy.asarray(src), numpy.asarray(snk)


BOUNDARY = {
    "stawiaski": lambda lab, args: stawiaski_edges(lab, args),
    "stawiaski_directed": lambda lab, args: stawiaski_directed_edges(lab, args[0], args[1]),
    "difference_of_means": lambda lab, args: difference_of_means_edges(lab, args),
}


def build_label_graph(label_image, fg, bg, boundary=None, boundary_args=None, regional_args=None, kind=None):
    """graph_from_labels (generate.py:177-338) on the BK oracle, not yet solved"""
    from . import bk
    label_image = numpy.asarray(label_image)
    fg = numpy.asarray(fg, dtype=numpy.bool_)
    bg = numpy.asarray(bg, dtype=numpy.bool_)
    nodes = len(numpy.unique(label_image))
    g = bk.BKGraph(nodes, 10 * nodes, kind)  # generate.py:296-306
    if regional_args is not None:  # generate.py:322-324
        src, snk = atlas_tweights(label_image, regional_args[0], regional_args[1])
        g.add_tweights(None, src, snk)
    if boundary is not None:  # generate.py:329
        i, j, cap, rev, once = BOUNDARY[boundary](label_image, boundary_args)
        if once.any():  # one call per distinct pair
            key = i * nodes + j
            _, first = numpy.unique(key, return_index=True)
            first.sort()
            i, j, cap, rev = i[first], j[first], cap[first], rev[first]
        g.sum_edges(i, j, cap, rev)
    s = numpy.unique(label_image[fg] - 1)  # generate.py:335-338
    g.add_tweights(s, numpy.full(s.size, 65535.0), numpy.zeros(s.size))
    t = numpy.unique(label_image[bg] - 1)
    g.add_tweights(t, numpy.zeros(t.size), numpy.full(t.size, 65535.0))
    return g


def graphcut_labels(label_image, fg, bg, boundary=None, boundary_args=None, regional_args=None, kind=None):
    """graph_from_labels + maxflow.  Returns (flow, region labels[bool], solved graph)."""
    g = build_label_graph(label_image, fg, bg, boundary, boundary_args, regional_args, kind)
    flow = g.maxflow()
    return flow, g.labels().astype(numpy.bool_), g
