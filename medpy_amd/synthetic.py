"""Deterministic synthetic volumes for the parity tests and bench.py.

No reference counterpart: MedPy ships no benchmark inputs.  Shapes are (Z, Y, X),
C-contiguous, so the node id of voxel (z, y, x) is (z*Y + y)*X + x exactly as the reference
numbers them (energy_voxel.py:667-677).
"""
import numpy as np


def _radius(shape):
    grids = np.ogrid[tuple(slice(0, s) for s in shape)]
    r2 = sum((g - (s - 1) / 2.0) ** 2 for g, s in zip(grids, shape))
    return np.sqrt(r2)


def _faces(shape):
    bg = np.zeros(shape, dtype=np.bool_)
    for d in range(len(shape)):
        sl = [slice(None)] * len(shape)
        sl[d] = 0
        bg[tuple(sl)] = True
        sl[d] = -1
        bg[tuple(sl)] = True
    return bg


def sphere(shape, step=100.0, noise=10.0, seed=0):
    """Bright ball (r < 0.3 n) in Gaussian noise; fg = inner ball r < 0.1 n; bg = the 6 faces; sigma 15."""
    shape = tuple(int(s) for s in shape)
    n = min(shape)
    r = _radius(shape)
    img = (step * (r < 0.3 * n)).astype(np.float32)
    img += np.random.default_rng(seed).normal(0.0, noise, shape).astype(np.float32)
    fg = r < 0.1 * n
    bg = _faces(shape)
    return {"image": img, "fg": fg, "bg": bg, "sigma": 15.0, "term": "difference_exponential"}


def hard(shape, seed=0):
    """Low-contrast stress variant (step 30, noise 15)."""
    return sphere(shape, step=30.0, noise=15.0, seed=seed)


def ties(shape, seed=0):
    """Tie-heavy parity stress: intensities in {0..3}, sigma 1, 1 % fg / 1 % bg random markers."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 4, shape).astype(np.float32)
    u = rng.random(shape)
    fg = u < 0.01
    bg = (u >= 0.01) & (u < 0.02)
    return {"image": img, "fg": fg, "bg": bg, "sigma": 1.0, "term": "difference_exponential"}


def ct(shape, seed=0):
    """Integer-valued volume in the manner of CT / MR data (uint16): soft tissue at 1040 +- 20 around a denser organ (1100 +- 20,
    r < 0.3 n), whole numbers throughout -- so the exponential term goes by table (graph.py:boundary_table), weights repeat
    everywhere and exact ties between cuts are the rule; sigma 25; fg = r < 0.1 n, bg = the six faces."""
    shape = tuple(int(s) for s in shape)
    n = min(shape)
    r = _radius(shape)
    img = 1040.0 + 60.0 * (r < 0.3 * n) + np.random.default_rng(seed).normal(0.0, 20.0, shape)
    img = np.clip(np.rint(img), 0, 65535).astype(np.uint16)
    return {"image": img, "fg": r < 0.1 * n, "bg": _faces(shape), "sigma": 25.0, "term": "difference_exponential"}


def regional(shape, seed=1):
    """float32 probability map for regional_probability_map (alpha 0.5)."""
    shape = tuple(int(s) for s in shape)
    n = min(shape)
    r = _radius(shape)
    p = 0.3 + 0.4 * (r < 0.3 * n) + np.random.default_rng(seed).normal(0.0, 0.1, shape)
    return {"prob": np.clip(p, 0.0, 1.0).astype(np.float32), "alpha": 0.5}
