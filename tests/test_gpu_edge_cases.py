"""-m gpu: empty / degenerate inputs the reference tolerates: no markers, one-sided markers, single
voxels, singleton axes, 1-D volumes, repeated solves.  Labels against the BK oracle."""
import numpy as np
import pytest

from oracle import pipeline

pytestmark = pytest.mark.gpu


def _cut(image, fg, bg, term="difference_exponential", sigma=5.0):
    from medpy_amd import graphcut
    args = (image, False) if term.endswith("linear") else (image, sigma, False)
    g = graphcut.graph_from_voxels(fg, bg, boundary_term=getattr(graphcut.energy_voxel, "boundary_" + term), boundary_term_args=args)
    flow = g.maxflow()
    ref = pipeline.graphcut_voxel(fg, bg, term=term, image=image, sigma=sigma)
    np.testing.assert_array_equal(g.labels(), ref.labels)
    assert flow == pytest.approx(ref.flow, rel=1e-9, abs=1e-300)
    return g, flow


@pytest.mark.parametrize("shape", [(1,), (7,), (1, 1), (1, 9), (9, 1), (3, 1, 5), (1, 1, 1), (1, 1, 17), (2, 2, 2), (8, 8, 8), (9, 8, 7)])
def test_small_and_singleton_shapes(shape):
    rng = np.random.default_rng(sum(shape))
    image = rng.normal(0, 10, shape).astype(np.float32)
    fg = np.zeros(shape, bool); bg = np.zeros(shape, bool)
    fg.flat[0] = True
    if fg.size > 1:
        bg.flat[-1] = True
    _cut(image, fg, bg)


def test_no_markers_at_all():
    """reference generate.py:169-172 skips empty marker sets; with no terminals every node is free -> label 1"""
    image = np.random.default_rng(0).normal(0, 10, (6, 7, 8)).astype(np.float32)
    z = np.zeros(image.shape, bool)
    g, flow = _cut(image, z, z)
    assert g.labels().all() and flow == 0.0


def test_only_foreground_or_only_background():
    image = np.random.default_rng(1).normal(0, 10, (10, 9, 12)).astype(np.float32)
    z = np.zeros(image.shape, bool)
    m = np.zeros(image.shape, bool); m[2:4, 3:5, 1:6] = True
    g, _ = _cut(image, m, z)
    assert g.labels().all()          # no sink: nobody can reach it
    g, _ = _cut(image, z, m)
    assert not g.labels().any()      # no source: everything connected to the sink markers is sink side


def test_everything_marked():
    image = np.random.default_rng(2).normal(0, 10, (5, 6, 7)).astype(np.float32)
    fg = np.ones(image.shape, bool); bg = np.zeros(image.shape, bool)
    _cut(image, fg, bg)
    _cut(image, bg, fg)
    _cut(image, fg, fg)  # every voxel both: all t-links cancel (graph.h:416-425)


def test_repeated_solve_and_rebuild_are_idempotent():
    from medpy_amd import graphcut, synthetic
    s = synthetic.sphere((24, 24, 24))
    g = graphcut.graph_from_voxels(s["fg"], s["bg"], boundary_term=graphcut.energy_voxel.boundary_difference_exponential,
                                   boundary_term_args=(s["image"], s["sigma"], False))
    f1 = g.maxflow(); l1 = g.labels().copy()
    f2 = g.maxflow()
    assert f1 == f2
    g._build()
    f3 = g.maxflow()
    assert f3 == f1 and np.array_equal(g.labels(), l1)
    assert g.what_segment(0) == g.termtype.SINK and int(g.what_segment(int(np.flatnonzero(s["fg"].ravel())[0]))) == 0


def test_constant_image_and_huge_dynamic_range():
    """all weights equal (exp(0) = 1) -> massive ties with exact arithmetic; and weights spanning DBL_MIN..1"""
    shape = (12, 12, 12)
    fg = np.zeros(shape, bool); fg[5:7, 5:7, 5:7] = True
    bg = np.zeros(shape, bool); bg[0] = True
    _cut(np.zeros(shape, np.float32), fg, bg)
    image = np.zeros(shape, np.float64); image[:, :, 6:] = 1e6  # exp(-1e12/..) underflows -> DBL_MIN clamp (energy_voxel.py:299)
    _cut(image, fg, bg, sigma=1.0)
