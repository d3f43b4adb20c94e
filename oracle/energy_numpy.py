"""NumPy restatement of the reference's voxel energy terms.  TEST INFRASTRUCTURE ONLY.

Follows reference medpy/graphcut/energy_voxel.py operation by operation (same ufuncs, same
order, same dtypes) but returns whole arrays instead of driving ``GCGraph.set_nweight`` once
per edge (energy_voxel.py:660-664).  Pinned against the reference itself (run through
``oracle/overlay.py`` with a recording graph) by ``tests/test_oracle_golden.py`` (where /root/reference exists) and
by the fixtures under ``tests/golden/``.

``boundary_weights`` returns, for every axis ``d``, the array the reference calls
``neighbourhood_intensity_term`` after the optional spacing division
(energy_voxel.py:644-658): shape = image.shape with ``shape[d]-1`` along ``d``; element
``[.., k, ..]`` is the symmetric capacity of the n-link between voxel ``k`` and ``k+1``
along ``d``.

The 26-neighbourhood has no reference implementation (generate.py:44-49 supports only
2*ndim); ``boundary_weights_offsets`` is this repository's definition of it.
"""
import math
import sys

import numpy

DBL_MIN = sys.float_info.min  # energy_voxel.py:113,188,235,299,344,406,451,513

TERMS = (
    "difference_linear", "difference_exponential", "difference_division", "difference_power",
    "maximum_linear", "maximum_exponential", "maximum_division", "maximum_power",
)


def _g_linear(x, m):  # energy_voxel.py:103-114 / 178-189
    x /= m
    x = 1.0 - x
    x[x == 0.0] = DBL_MIN
    return x


def _g_exponential(x, sigma):  # energy_voxel.py:226-236 / 290-300
    x = numpy.power(x, 2)
    x /= math.pow(sigma, 2)
    x *= -1
    x = numpy.exp(x)
    x[x <= 0] = DBL_MIN
    return x


def _g_division(x, sigma):  # energy_voxel.py:337-345 / 399-407
    x /= sigma
    x = 1.0 / (x + 1)
    x[x <= 0] = DBL_MIN
    return x


def _g_power(x, sigma):  # energy_voxel.py:444-452 / 506-514
    x = 1.0 / (x + 1)
    x = numpy.power(x, sigma)
    x[x <= 0] = DBL_MIN
    return x


def _prepare(term, image, sigma):
    """Returns (float64 image, neighbourhood function, g) exactly as the reference composes them."""
    image = numpy.asarray(image)
    family, kind = term.split("_")
    if kind == "linear":
        if family == "maximum":
            m = float(numpy.abs(image).max())  # energy_voxel.py:101
        else:
            m = float(abs(image.max() - image.min()))  # energy_voxel.py:174-176
        g = lambda x: _g_linear(x, m)
    elif kind == "exponential":
        g = lambda x: _g_exponential(x, sigma)
    elif kind == "division":
        g = lambda x: _g_division(x, sigma)
    elif kind == "power":
        g = lambda x: _g_power(x, sigma)
    else:
        raise ValueError(term)
    # boundary_maximum_division calls __skeleton_difference (energy_voxel.py:347): a reference
    # quirk that parity must keep.
    use_max = family == "maximum" and kind != "division"
    if use_max:
        image = numpy.abs(image)  # energy_voxel.py:558
        nb = lambda a, b: numpy.maximum(a, b)  # energy_voxel.py:556
    else:
        nb = lambda a, b: numpy.absolute(a - b)  # energy_voxel.py:606
    image = image.astype(float)  # energy_voxel.py:634
    return image, nb, g


def boundary_weights(term, image, sigma=None, spacing=False):
    """Per-axis n-link weights of __skeleton_base (energy_voxel.py:611-664), 2*ndim neighbourhood."""
    image, nb, g = _prepare(term, image, sigma)
    out = []
    for dim in range(image.ndim):
        lo = [slice(None)] * image.ndim
        hi = [slice(None)] * image.ndim
        lo[dim] = slice(-1)
        hi[dim] = slice(1, None)
        w = g(nb(image[tuple(lo)], image[tuple(hi)]))
        if spacing:
            w /= spacing[dim]  # energy_voxel.py:657-658, after the clamp
        out.append(numpy.ascontiguousarray(w))
    return out


def forward_offsets(ndim, connectivity):
    """Lexicographically positive neighbour offsets.  connectivity = 2*ndim (reference) or 3**ndim-1."""
    if connectivity == 2 * ndim:
        return [tuple(1 if k == d else 0 for k in range(ndim)) for d in range(ndim)]
    if connectivity != 3 ** ndim - 1:
        raise ValueError("connectivity must be %d or %d" % (2 * ndim, 3 ** ndim - 1))
    offs = []
    for code in range(3 ** ndim):
        o = tuple((code // 3 ** (ndim - 1 - k)) % 3 - 1 for k in range(ndim))
        if o > tuple([0] * ndim):
            offs.append(o)
    return offs


def boundary_weights_offsets(term, image, offsets, sigma=None, spacing=False):
    """Extension (no reference counterpart): the same g() on arbitrary neighbour offsets.

    w = g(nb(I_p, I_{p+o})); with spacing the weight is divided by the Euclidean length of
    (o * spacing), which reduces to ``/ spacing[d]`` on the axes (energy_voxel.py:657-658).
    Returns {offset: array of shape image.shape (NaN where p+o falls outside)}.
    """
    image, nb, g = _prepare(term, image, sigma)
    res = {}
    for o in offsets:
        src, dst = [], []
        for k, ok in enumerate(o):
            n = image.shape[k]
            if ok == 0:
                src.append(slice(0, n)); dst.append(slice(0, n))
            elif ok > 0:
                src.append(slice(0, n - 1)); dst.append(slice(1, n))
            else:
                src.append(slice(1, n)); dst.append(slice(0, n - 1))
        w = g(nb(image[tuple(src)], image[tuple(dst)]))
        if spacing:
            w /= math.sqrt(sum((ok * s) ** 2 for ok, s in zip(o, spacing)))
        full = numpy.full(image.shape, numpy.nan)
        full[tuple(src)] = w
        res[tuple(o)] = full
    return res


def regional_probability_tweights(probability_map, alpha):
    """(source, sink) t-link capacities of regional_probability_map (energy_voxel.py:61-65).

    Evaluated in the dtype of ``probability_map`` (float32 map -> float32 products), then
    widened by ``float()`` in GCGraph.set_tweight (graph.py:496-498).
    """
    pm = numpy.asarray(probability_map)
    src = numpy.asarray((pm * alpha)).ravel().astype(numpy.float64)
    snk = numpy.asarray(((1 - pm) * alpha)).ravel().astype(numpy.float64)
    return src, snk
