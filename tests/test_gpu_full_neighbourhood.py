"""-m gpu: the full neighbourhood (26 in 3-D, 8 in 2-D) -- BASELINE.json configs 3 and 5.  The reference has
no such graph (generate.py:44-49), so the oracle is the reference BK core fed the edge list of
oracle/energy_numpy.py:boundary_weights_offsets: labels bit-exact, energies <= 1e-6 (bitwise for
the terms made of IEEE basic operations), flow relative 1e-9."""
import numpy as np
import pytest

from oracle import energy_numpy, pipeline

pytestmark = pytest.mark.gpu


def _run(s, conn, term=None, sigma=None, spacing=False, prob=None, alpha=None):
    from medpy_amd import graphcut
    term = term or s["term"]
    sigma = s["sigma"] if sigma is None else sigma
    kw = dict(boundary_term=getattr(graphcut.energy_voxel, "boundary_" + term),
              boundary_term_args=(s["image"], spacing) if term.endswith("linear") else (s["image"], sigma, spacing))
    if prob is not None:
        kw.update(regional_term=graphcut.energy_voxel.regional_probability_map, regional_term_args=(prob, alpha))
    g = graphcut.graph_from_voxels(s["fg"], s["bg"], connectivity=conn, **kw)
    flow = g.maxflow()
    return g, flow


def _check(s, conn, term=None, sigma=None, spacing=False, prob=None, alpha=None, exact_energy=False):
    term = term or s["term"]
    sigma = s["sigma"] if sigma is None else sigma
    g, flow = _run(s, conn, term, sigma, spacing, prob, alpha)
    nd = s["image"].ndim
    offs = energy_numpy.forward_offsets(nd, conn)
    wref = energy_numpy.boundary_weights_offsets(term, s["image"], offs, sigma, spacing)
    wdev = {}
    for o in offs:
        wdev[o] = g.nweights_offset(o)
        a, b = wdev[o], wref[o]
        assert np.array_equal(np.isnan(a), np.isnan(b))
        if exact_energy:
            np.testing.assert_array_equal(a, b)
        else:
            assert np.nanmax(np.abs(a - b)) <= 1e-6
        # the reverse arc carries the same capacity
        rev = g.nweights_offset(tuple(-k for k in o))
        src = tuple(slice(max(0, -k), a.shape[i] - max(0, k)) for i, k in enumerate(o))
        dst = tuple(slice(max(0, k), a.shape[i] - max(0, -k)) for i, k in enumerate(o))
        np.testing.assert_array_equal(rev[dst], a[src])
    kw = dict(prob=prob, alpha=alpha) if prob is not None else {}
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term=term, image=s["image"], sigma=sigma, spacing=spacing, connectivity=conn, **kw)
    inj = pipeline.graphcut_voxel(s["fg"], s["bg"], weights=wdev, connectivity=conn, **kw)
    np.testing.assert_array_equal(g.labels(), inj.labels)
    np.testing.assert_array_equal(g.labels(), ref.labels)
    assert flow == pytest.approx(ref.flow, rel=1e-9)
    print("conn %d %s %s: flow %.12g, fg %.4f, %s" % (conn, term, s["image"].shape, flow, g.labels().mean(), g.stats()))
    return g


@pytest.mark.parametrize("gen,shape", [("sphere", (32, 32, 32)), ("hard", (48, 48, 48)), ("sphere", (20, 33, 47)), ("sphere", (64, 64, 64))])
def test_26_vs_oracle(gen, shape):
    from medpy_amd import synthetic
    _check(getattr(synthetic, gen)(shape), 26)


def test_8_neighbourhood_2d():
    from medpy_amd import synthetic
    s3 = synthetic.sphere((1, 48, 72))
    s = {k: (v[0] if isinstance(v, np.ndarray) else v) for k, v in s3.items()}
    s["bg"] = np.zeros_like(s["fg"]); s["bg"][0] = s["bg"][-1] = True; s["bg"][:, 0] = s["bg"][:, -1] = True
    _check(s, 8)


def test_26_spacing_and_exact_terms():
    from medpy_amd import synthetic
    s = synthetic.sphere((24, 28, 20))
    _check(s, 26, term="difference_division", sigma=3.0, spacing=(1.0, 2.0, 0.5), exact_energy=True)
    _check(s, 26, term="maximum_linear", exact_energy=True)


def test_config3_26conn_plus_regional():
    """BASELINE.json configs[2] at the size the CPU oracle reaches: 26-conn + regional_probability_map t-links."""
    from medpy_amd import synthetic
    shape = (96, 96, 96)
    s = synthetic.sphere(shape)
    r = synthetic.regional(shape)
    g = _check(s, 26, prob=r["prob"], alpha=r["alpha"])
    src, snk = energy_numpy.regional_probability_tweights(r["prob"], r["alpha"])
    tr = g.tweights().ravel()
    free = ~(s["fg"].ravel() | s["bg"].ravel())
    np.testing.assert_array_equal(tr[free], (src - snk)[free])  # float32 products, exact (energy_voxel.py:61-65)
