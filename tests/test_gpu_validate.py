"""-m gpu: mgc_validate -- the invariants of a maximum preflow counted on the device (the reference's
Graph::test_consistency, lib/maxflow/src/maxflow.cpp:610-682, in spirit).  It is the only check available for the
multi-GPU volumes no CPU oracle reaches (BASELINE.json configs 4 and 5), so it is pinned here where the oracle also
reaches: a correct solve has no violation and flow == cut; a solve cut short is reported as such."""
import numpy as np
import pytest

from oracle import pipeline

pytestmark = pytest.mark.gpu


def _graph(s, connectivity=None, regional=None):
    from medpy_amd import graphcut
    kw = dict(boundary_term=graphcut.energy_voxel.boundary_difference_exponential, boundary_term_args=(s["image"], s["sigma"], False))
    if regional is not None:
        kw.update(regional_term=graphcut.energy_voxel.regional_probability_map, regional_term_args=(regional["prob"], regional["alpha"]))
    if connectivity:
        kw["connectivity"] = connectivity
    return graphcut.graph_from_voxels(s["fg"], s["bg"], **kw)


@pytest.mark.parametrize("gen,shape,conn,reg", [("sphere", (64, 64, 64), None, False), ("hard", (48, 40, 56), None, False),
                                                ("ties", (40, 40, 40), None, False), ("sphere", (37, 21, 50), None, True),
                                                ("sphere", (48, 48, 48), 26, False), ("sphere", (40, 40, 40), 26, True)])
def test_a_correct_solve_has_no_violation(gen, shape, conn, reg):
    from medpy_amd import _lib, synthetic
    s = getattr(synthetic, gen)(shape)
    r = synthetic.regional(shape) if reg else None
    g = _graph(s, conn, r)
    flow = g.maxflow()
    v = g.validate()
    assert v["voxels"] == int(np.prod(shape))
    diff = _lib.assert_valid(v)
    assert v["max_pair_error"] <= 1e-9 and v["max_node_error"] <= 1e-9
    assert flow == pytest.approx(v["cut_capacity"] + v["flow_constant"], rel=1e-12)
    print(gen, shape, conn, "flow/cut rel diff %.2e, pair err %.2e, node err %.2e" % (diff, v["max_pair_error"], v["max_node_error"]))


def test_a_solve_cut_short_is_reported():
    from medpy_amd import _lib, synthetic
    s = synthetic.sphere((96, 96, 96))
    g = _graph(s)
    g.set_param("max_outer", 1)  # one global relabel, eight rounds: excess is still on its way
    with pytest.raises(_lib.MedpyHipError):
        g.maxflow()
    v = g.validate()
    assert v["active_excess"] > 0
    with pytest.raises(AssertionError, match="active_excess"):
        _lib.assert_valid(v)
    assert v["negative_values"] == 0  # a preflow all the same
    # (flow still travelling in an outbox has left one end of its arc pair and not reached the other yet: the conservation
    # counts are only meaningful once nothing is pending)
    assert v["pending_outbox"] > 0 or (v["pair_violations"] == 0 and v["node_violations"] == 0)
    g.set_param("max_outer", 100000)
    g._build()
    g.maxflow()
    _lib.assert_valid(g.validate())


@pytest.mark.parametrize("conn", [6, 26])
def test_slabs_validate_globally(conn):
    """the N > 1 form: every slab counts over its own planes, counts and flows are summed over the slabs"""
    from medpy_amd import _lib, synthetic
    from medpy_amd.slab import HipSlab, LoopbackExchange, solve_slabs, validate_slabs
    shape = (64, 40, 48)
    s = synthetic.sphere(shape)
    slabs = [HipSlab(shape, r, 3, connectivity=conn) for r in range(3)]
    for sl in slabs:
        z = slice(sl.plane0, sl.plane1)
        sl.set_boundary(s["term"], s["image"][z], s["sigma"])
        sl.set_markers(s["fg"][z], s["bg"][z])
        sl.build()
    ex = LoopbackExchange(slabs)
    st = solve_slabs(slabs, ex)
    assert st["converged"] == 1
    flow = sum(sl.finish()[1] for sl in slabs)
    v = validate_slabs(slabs, ex)
    assert v["voxels"] == int(np.prod(shape))
    _lib.assert_valid(v)
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term=s["term"], image=s["image"], sigma=s["sigma"],
                                  connectivity=conn if conn != 6 else None)
    assert flow == pytest.approx(ref.flow, rel=1e-9)
    assert v["cut_capacity"] + v["flow_constant"] == pytest.approx(ref.flow, rel=1e-9)
    for sl in slabs:
        sl.close()
