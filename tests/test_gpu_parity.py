"""-m gpu: the HIP path (through the C ABI, via the Python host layer) against
 (a) the reference's own outputs committed under tests/golden/ and
 (b) the CPU oracle (compiled reference BK when it travelled, else the C restatement) on the
     same seeded inputs.
Bars: labels bit-exact; t-links bit-exact; n-link energies bit-exact for the terms made of
IEEE basic operations (linear / division), |delta| <= 1e-6 for exp / pow (libm vs OCML);
flow relative 1e-9.

Tie-degenerate inputs (integer-valued images, the maximum_* terms): several minimum cuts exist EXACTLY and which one a
floating point solver reports hinges on its rounding history.  There the tests do not accept "a few" differing voxels:
they require (oracle/cutcheck.py) that every differing voxel lies in the AMBIGUITY SET of the reference's own residual
graph (neither reachable from the source nor able to reach the sink once residuals of a few ulp count as saturated),
and that the two cuts have the same capacity when each is evaluated in exact rational arithmetic."""
import numpy as np
import pytest

from oracle import bk, cutcheck, energy_numpy, pipeline

pytestmark = pytest.mark.gpu

TERMS = energy_numpy.TERMS
ENERGY_TOL = 1e-6


def _gc():
    from medpy_amd import graphcut
    return graphcut


def _term_fn(term):
    return getattr(_gc().energy_voxel, "boundary_" + term)


def _term_args(term, image, sigma, spacing):
    return (image, spacing) if term.endswith("linear") else (image, sigma, spacing)


def _run(fg, bg, term=None, image=None, sigma=None, spacing=False, prob=None, alpha=None):
    gc = _gc()
    kw = {}
    if term is not None:
        kw["boundary_term"] = _term_fn(term)
        kw["boundary_term_args"] = _term_args(term, image, sigma, spacing)
    if prob is not None:
        kw["regional_term"] = gc.energy_voxel.regional_probability_map
        kw["regional_term_args"] = (prob, alpha)
    g = gc.graph_from_voxels(fg, bg, **kw)
    return g


def _edge_weights_from_graph(g, shape, ei, ej):
    """weights of the golden edge list (i<j lattice neighbours) from the per-axis read-back"""
    nd = len(shape)
    strides = [int(np.prod(shape[k + 1:])) for k in range(nd)]
    out = np.empty(ei.size)
    axes = [g.nweights(a) for a in range(nd)]
    for k, (i, j) in enumerate(zip(ei, ej)):
        a = strides.index(int(j - i))
        idx = np.unravel_index(int(i), shape)
        out[k] = axes[a][idx]
    return out


def _check_energy(term, got, want):
    if term.split("_")[1] in ("linear", "division"):
        np.testing.assert_array_equal(got, want)
    else:
        assert np.max(np.abs(got - want), initial=0.0) <= ENERGY_TOL
        # report how close to bitwise we are (informational)
        ulp = np.abs(got - want) / np.maximum(np.spacing(np.abs(want)), 1e-320)
        print("%s: max |delta| %.3e, max ulp %.1f, bitwise-equal fraction %.4f" % (term, np.max(np.abs(got - want), initial=0.0),
                                                                                   ulp.max(initial=0.0), float(np.mean(got == want))))


def test_reference_cut_kat(golden_kat):
    """reference tests/graphcut_/cut.py:32-50: 2x3x5 volume, difference_linear: labels and maxflow == 3."""
    g0 = golden_kat.group("cut")
    g = _run(g0["fg"], g0["bg"], "difference_linear", g0["image"])
    flow = g.maxflow()
    np.testing.assert_array_equal(g.labels(), g0["labels"].astype(bool))
    assert flow == pytest.approx(3.0, rel=1e-12)
    # the reference's own read-out loop (bin/medpy_graphcut_voxel.py:177-181)
    res = np.array([0 if g.termtype.SINK == g.what_segment(i) else 1 for i in range(g0["image"].size)])
    np.testing.assert_array_equal(res.reshape(g0["image"].shape), g0["labels"])


@pytest.mark.parametrize("term", TERMS)
def test_reference_energy_2d_kat(golden_kat, term):
    """reference tests/graphcut_/energy_voxel.py:55-103."""
    top = golden_kat.group("e2d")
    g0 = golden_kat.group("e2d/" + term)
    image = top["image"] if term.startswith("difference") else top["gradient"]
    sigma = {"exponential": 1.0, "division": 0.5, "power": 2.0, "linear": None}[term.split("_")[1]]
    g = _run(top["fg"], top["bg"], term, image, sigma)
    _check_energy(term, _edge_weights_from_graph(g, image.shape, g0["edges_i"], g0["edges_j"]), g0["edges_w"])
    flow = g.maxflow()
    np.testing.assert_array_equal(g.labels(), g0["labels"].astype(bool))
    assert flow == pytest.approx(float(g0["flow"]), rel=1e-9)


def test_reference_regional_and_spacing_kat(golden_kat):
    top = golden_kat.group("e2d")
    g0 = golden_kat.group("e2d/regional")
    g = _run(top["fg"], top["bg"], prob=top["image"] / 2.0, alpha=1.0)
    np.testing.assert_array_equal(g.tweights().ravel(), g0["trcap"])
    flow = g.maxflow()
    np.testing.assert_array_equal(g.labels(), g0["labels"].astype(bool))
    assert flow == pytest.approx(float(g0["flow"]), rel=1e-9)
    s = golden_kat.group("spacing")
    g = _run(s["fg"], s["bg"], "difference_division", s["image"], 1.0, (1.0, 5.0))
    np.testing.assert_array_equal(_edge_weights_from_graph(g, s["image"].shape, s["edges_i"], s["edges_j"]), s["edges_w"])
    g.maxflow()
    np.testing.assert_array_equal(g.labels(), s["labels"].astype(bool))


@pytest.mark.parametrize("case", ["c0", "c1", "c2", "c3", "c5"])
@pytest.mark.parametrize("term", TERMS)
def test_small_volumes_all_terms(golden_small, case, term):
    top = golden_small.group(case)
    g0 = golden_small.group("%s/%s" % (case, term))
    spacing = tuple(top["spacing"]) if top["spacing"].size else False
    sigma = float(top["sigma"])
    g = _run(top["fg"], top["bg"], term, top["image"], sigma, spacing)
    _check_energy(term, _edge_weights_from_graph(g, top["image"].shape, g0["edges_i"], g0["edges_j"]), g0["edges_w"])
    np.testing.assert_array_equal(g.tweights().ravel(), g0["trcap"])
    flow = g.maxflow()
    assert flow == pytest.approx(float(g0["flow"]), rel=1e-9)  # the cut found IS a minimum cut
    labels = g.labels()
    if (labels != g0["labels"].astype(bool)).any():
        assert _tie_degenerate(term, top["image"]), "labels differ on an input without exact ties"
        _assert_equivalent_to_reference(labels, top, term, sigma, spacing, g0)


def _assert_equivalent_to_reference(labels, top, term, sigma, spacing, g0, max_differing=None):
    """the oracle re-solves the case (its labels are the fixture's, checked), then every differing voxel must be ambiguous
    in ITS residual graph and the two cuts must cost exactly the same (rational arithmetic)"""
    ref = pipeline.graphcut_voxel(top["fg"], top["bg"], term=term, image=top["image"], sigma=sigma, spacing=spacing)
    np.testing.assert_array_equal(ref.labels, g0["labels"].astype(bool))
    w = energy_numpy.boundary_weights(term, top["image"], sigma, spacing)
    i, j, ww = cutcheck.lattice_edges(top["image"].shape, w)
    nbad = cutcheck.assert_labels_equivalent(labels, ref, max_differing, exact=(i, j, ww, ww, g0["trcap"]))
    print("tie-degenerate case: %d ambiguous voxel(s) labelled differently, cut capacities exactly equal" % nbad)


def _tie_degenerate(term, image):
    """maximum_* terms give every edge of a local-maximum voxel the same weight, integer images repeat
    weights everywhere: exact ties between cuts, resolved by floating point rounding order."""
    return term.startswith("maximum") or np.issubdtype(np.asarray(image).dtype, np.integer)


@pytest.mark.parametrize("term", TERMS)
def test_4d_volume_against_the_reference(golden_small, term):
    """a 4-D image (ndim*2 = 8 neighbours, generate.py:44-49): routed to the sparse-graph solver, n-links generated in HBM
    for any number of axes; fixtures are the reference's own results"""
    top = golden_small.group("c4")
    g0 = golden_small.group("c4/%s" % term)
    spacing = tuple(top["spacing"]) if top["spacing"].size else False
    g = _run(top["fg"], top["bg"], term, top["image"], float(top["sigma"]), spacing)
    got = np.array([g.get_edge(int(i), int(j)) for i, j in zip(g0["edges_i"][::7], g0["edges_j"][::7])])
    _check_energy(term, got, g0["edges_w"][::7])
    np.testing.assert_array_equal(np.array([g.get_trcap(i) for i in range(top["fg"].size)]), g0["trcap"])
    flow = g.maxflow()
    assert flow == pytest.approx(float(g0["flow"]), rel=1e-9)
    labels = g.labels().reshape(top["fg"].shape)
    if (labels != g0["labels"].astype(bool)).any():
        assert _tie_degenerate(term, top["image"]), "labels differ on an input without exact ties"
        _assert_equivalent_to_reference(labels, top, term, float(top["sigma"]), spacing, g0)


@pytest.mark.parametrize("case", ["r0", "r1"])
def test_regional_tlink_merge(golden_small, case):
    top = golden_small.group(case)
    for sub, kw in (("cut", dict(term="difference_exponential", image=top["image"], sigma=float(top["sigma"]))),
                    ("regional_only", {})):
        g0 = golden_small.group("%s/%s" % (case, sub))
        g = _run(top["fg"], top["bg"], prob=top["prob"], alpha=float(top["alpha"]), **kw)
        np.testing.assert_array_equal(g.tweights().ravel(), g0["trcap"])  # bit-exact t-links incl. fg&bg voxel
        flow = g.maxflow()
        np.testing.assert_array_equal(g.labels(), g0["labels"].astype(bool))
        assert flow == pytest.approx(float(g0["flow"]), rel=1e-9)


def test_golden_synthetic(golden_synth):
    from medpy_amd import synthetic
    for key in sorted({k.split("/")[0] for k in golden_synth._z.files}):
        gen, dims = key.rsplit("_", 1)
        shape = tuple(int(x) for x in dims.split("x"))
        s = getattr(synthetic, gen)(shape)
        g = _run(s["fg"], s["bg"], s["term"], s["image"], s["sigma"])
        flow = g.maxflow()
        lab = np.unpackbits(golden_synth[key + "/labels"])[: int(np.prod(shape))].reshape(shape).astype(bool)
        labels = g.labels()
        if (labels != lab).any():
            assert gen == "ties", "labels differ on an input without exact ties"
            ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term=s["term"], image=s["image"], sigma=s["sigma"])
            np.testing.assert_array_equal(ref.labels, lab)
            w = energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"])
            i, j, ww = cutcheck.lattice_edges(shape, w)
            tr = np.where(s["fg"], 65535.0, 0.0) - np.where(s["bg"], 65535.0, 0.0)
            cutcheck.assert_labels_equivalent(labels, ref, exact=(i, j, ww, ww, tr))
        assert flow == pytest.approx(float(golden_synth[key + "/flow"]), rel=1e-9)


def _oracle_vs_gpu(gen, shape, regional=False):
    from medpy_amd import synthetic
    s = getattr(synthetic, gen)(shape)
    kw = {}
    if regional:
        r = synthetic.regional(shape)
        kw = dict(prob=r["prob"], alpha=r["alpha"])
    g = _run(s["fg"], s["bg"], s["term"], s["image"], s["sigma"], **kw)
    flow = g.maxflow()
    labels = g.labels()
    # (1) whole pipeline against the oracle (NumPy weights + BK)
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term=s["term"], image=s["image"], sigma=s["sigma"], **kw)
    # (2) cross-inject: device-built capacities into the CPU solver isolates the solve from exp() rounding
    wdev = [g.nweights(a) for a in range(3)]
    inj = pipeline.graphcut_voxel(s["fg"], s["bg"], weights=wdev, **kw)
    for a in range(3):
        wref = energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"])[a]
        assert np.max(np.abs(wdev[a] - wref)) <= ENERGY_TOL
    print("%s %s: flow gpu %.15g oracle %.15g, fg %.5f, stats %s" % (gen, shape, flow, ref.flow, labels.mean(), g.stats()))
    return labels, flow, ref, inj


@pytest.mark.parametrize("gen,shape", [("sphere", (32, 32, 32)), ("sphere", (64, 64, 64)), ("hard", (64, 64, 64)),
                                       ("sphere", (20, 33, 47)), ("sphere", (128, 128, 128)), ("hard", (96, 96, 96))])
def test_synthetic_vs_oracle(gen, shape):
    labels, flow, ref, inj = _oracle_vs_gpu(gen, shape)
    np.testing.assert_array_equal(labels, inj.labels)
    np.testing.assert_array_equal(labels, ref.labels)
    assert flow == pytest.approx(ref.flow, rel=1e-9)
    assert flow == pytest.approx(inj.flow, rel=1e-9)


@pytest.mark.parametrize("radial", [0, 1, 2])
@pytest.mark.parametrize("gen,shape,kw", [("sphere", (64, 64, 64), {}), ("sphere", (96, 128, 80), {}), ("hard", (96, 96, 96), {}), ("ties", (48, 48, 48), {}),
                                          ("sphere", (128, 128, 128), dict(radial_budget_x16=3)), ("sphere", (128, 128, 128), dict(radial_rounds0=2)),
                                          ("sphere", (40, 40, 40), dict(radial_min_c=1, radial_min_walls=0))])
def test_flood_phase_on_radial_labels_reaches_the_same_cut(gen, shape, kw, radial):
    """mgc_dt_ops.inl / mgc_driver.inl, round 5: whether the flood phase of a solve runs on exact labels (radial = 0), on radial labels
    (1: whenever the first relabel is the distance transform) or decides by the walls k_build counted (2, the default) -- and however the
    radial phase is cut (a budget too short to close the surface, several radial cycles with a look at the source tiles in between,
    sources next to sinks) -- the maximum preflow differs and the labels do not: the reference BK's, voxel for voxel."""
    from medpy_amd import graphcut, synthetic
    s = getattr(synthetic, gen)(shape)
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term=s["term"], image=s["image"], sigma=s["sigma"])
    g = graphcut.graph_from_voxels(s["fg"], s["bg"], boundary_term=getattr(graphcut.energy_voxel, "boundary_" + s["term"]),
                                   boundary_term_args=(s["image"], s["sigma"], False))
    g.set_param("radial", radial)
    for k, v in kw.items():
        g.set_param(k, v)
    flow = g.maxflow()
    st = g.stats()
    print(gen, shape, kw, "radial", radial, "->", {k: st[k] for k in ("global_relabels", "phases", "discharge_tiles", "relabel_tiles")}, "radial cycles", st["radial_cycles"])
    if radial == 0:
        assert st["radial_cycles"] == 0
    if radial == 1 and gen == "sphere" and not kw.get("radial_budget_x16"):
        assert st["radial_cycles"] >= 1
    if gen == "ties":  # exact ties between minimum cuts: equivalent, not identical (oracle/cutcheck.py)
        i, j, ww = cutcheck.lattice_edges(shape, energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"]))
        tr = np.where(s["fg"], 65535.0, 0.0) - np.where(s["bg"], 65535.0, 0.0)
        cutcheck.assert_labels_equivalent(g.labels(), ref, exact=(i, j, ww, ww, tr))
    else:
        np.testing.assert_array_equal(g.labels(), ref.labels)
    assert flow == pytest.approx(ref.flow, rel=1e-9)
    v = g.validate()
    assert not any(v[k] for k in ("negative_values", "active_excess", "residual_arcs_across", "sink_links_across", "pair_violations", "node_violations", "pending_outbox"))


@pytest.mark.parametrize("gen,shape", [("sphere", (128, 128, 128)), ("hard", (96, 96, 96)), ("sphere", (20, 33, 47))])
def test_cpu_capacities_into_the_gpu_solver(gen, shape):
    """SURVEY Appendix B "cross-inject", the direction round 2 lacked at scale: the ORACLE's capacities (NumPy energies,
    t-links merged by the reference BK's add_tweights) go into the device solver through the plug-in path (mgc_add_edges +
    mgc_set_tweights_merged, no boundary term on the device), so the solve runs on bit-identical inputs: labels must equal
    the reference BK's voxel for voxel, whatever exp() the device would have used."""
    from medpy_amd import synthetic
    from medpy_amd.graphcut.graph import VoxelGraph
    s = getattr(synthetic, gen)(shape)
    w = energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"])
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], weights=w)
    i, j, ww = cutcheck.lattice_edges(shape, w)
    assert not (s["fg"] & s["bg"]).any()
    tr = np.where(s["fg"], 65535.0, 0.0) - np.where(s["bg"], 65535.0, 0.0)  # add_tweights on disjoint markers, graph.h:416-425
    g = VoxelGraph(shape)
    g._add_edges(i, j, ww, ww)
    g._set_tweights_merged(tr, 0.0)
    g._build()
    flow = g.maxflow()
    labels = g.labels()
    np.testing.assert_array_equal(labels, ref.labels)
    # flow = constant folded by add_tweights (zero here: disjoint markers) + capacity of the cut
    assert flow == pytest.approx(ref.flow, rel=1e-9)
    print("cpu capacities -> gpu solver %s %s: flow %.15g (oracle %.15g), stats %s" % (gen, shape, flow, ref.flow, g.stats()))


def test_regional_plus_boundary_vs_oracle():
    labels, flow, ref, inj = _oracle_vs_gpu("sphere", (48, 48, 48), regional=True)
    np.testing.assert_array_equal(labels, inj.labels)
    np.testing.assert_array_equal(labels, ref.labels)
    assert flow == pytest.approx(ref.flow, rel=1e-9)


def test_config2_256_cube_bit_exact():
    """BASELINE.json configs[1]: 256^3 synthetic volume, 6-conn, labels bit-exact vs the CPU reference."""
    labels, flow, ref, inj = _oracle_vs_gpu("sphere", (256, 256, 256))
    np.testing.assert_array_equal(labels, ref.labels)
    np.testing.assert_array_equal(labels, inj.labels)
    assert flow == pytest.approx(ref.flow, rel=1e-9)


def test_ties_dyadic_bit_exact():
    """Tie-heavy integer image with a term whose weights are dyadic (linear, range 4): all arithmetic is
    exact, so even degenerate cuts must match the reference solver voxel for voxel."""
    from medpy_amd import synthetic
    s = synthetic.ties((40, 40, 40))
    img = s["image"].copy()
    img.flat[0], img.flat[1] = 0.0, 4.0  # range exactly 4 -> weights in {1, .75, .5, .25, DBL_MIN}
    g = _run(s["fg"], s["bg"], "difference_linear", img)
    flow = g.maxflow()
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term="difference_linear", image=img)
    np.testing.assert_array_equal(g.labels(), ref.labels)
    assert flow == ref.flow


def test_ties_dyadic_bit_exact_in_the_wave_kernels_as_shipped():
    """The same at 96^3, where the colour phases are long enough for the wave kernels WITHOUT any test parameter, and where
    every tile holds a sink link: the library switches the exact in-tile labelling of such tiles on by itself
    (exact_sink_tiles = 1, decided from k_build's count).  Dyadic weights: bit-exact labels and flow, whatever the schedule."""
    from medpy_amd import synthetic
    s = synthetic.ties((96, 96, 96))
    img = s["image"].copy()
    img.flat[0], img.flat[1] = 0.0, 4.0
    g = _run(s["fg"], s["bg"], "difference_linear", img)
    flow = g.maxflow()
    st = g.stats()
    assert st["discharge_wave_tiles"] > 0, st  # k_discharge_w ran (not only the 512-thread form)
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term="difference_linear", image=img)
    np.testing.assert_array_equal(g.labels(), ref.labels)
    assert flow == ref.flow


def test_layouts_and_dtypes():
    """F-ordered / strided views as medpy.io.load returns them (io/load.py:127) and integer images."""
    from medpy_amd import synthetic
    s = synthetic.sphere((24, 30, 18))
    g = _run(s["fg"], s["bg"], s["term"], s["image"], s["sigma"])
    g.maxflow()
    base = g.labels()
    imgF, fgF, bgF = np.asfortranarray(s["image"]), np.asfortranarray(s["fg"]), np.asfortranarray(s["bg"])
    g2 = _run(fgF, bgF, s["term"], imgF, s["sigma"])
    g2.maxflow()
    np.testing.assert_array_equal(g2.labels(), base)
    img16 = np.clip(s["image"] + 50, 0, 1000).astype(np.uint16)
    g3 = _run(s["fg"], s["bg"], "difference_division", img16, 3.0)
    g3.maxflow()
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term="difference_division", image=img16, sigma=3.0)
    np.testing.assert_array_equal(g3.labels(), ref.labels)
    np.testing.assert_array_equal(g3.nweights(1), energy_numpy.boundary_weights("difference_division", img16, 3.0)[1])


def test_special_images_do_not_crash():
    """reference tests/graphcut_/energy_voxel.py:152-179 (negative / all-zero image, NaN weights): no crash, no hang."""
    fg = np.zeros((3, 3), bool); fg[2, 2] = True
    bg = np.zeros((3, 3), bool); bg[0, 0] = True
    for image in (np.asarray([[-1, 1, -4], [2, -7, 3], [-2.3, 3, -7]], dtype=float), np.zeros((3, 3))):
        for term in TERMS:
            g = _run(fg, bg, term, image, 1.0)
            g.maxflow()
            assert g.labels().shape == (3, 3)


def test_api_contract():
    gc = _gc()
    g = gc.GCGraph(6, 7, shape=(2, 3))
    assert g.get_node_count() == 6 and g.get_edge_count() == 7 and g.get_nodes() == list(range(6))
    with pytest.raises(ValueError):
        g.set_nweight(0, 6, 1.0, 1.0)
    with pytest.raises(ValueError):
        g.set_nweight(1, 1, 1.0, 1.0)
    with pytest.raises(ValueError):
        g.set_nweight(0, 1, 0.0, 1.0)
    with pytest.raises(ValueError):
        g.set_tweight(6, 1.0, 1.0)
    with pytest.raises(ValueError):
        g.set_source_nodes([7])
    # plug-in path: explicit lattice edges + explicit t-links through the reference-style setters
    g.set_nweight(0, 1, 2.0, 2.0); g.set_nweight(0, 1, 1.0, 0.5); g.set_nweight(1, 2, 1.0, 1.0)
    g.set_nweight(2, 5, 4.0, 4.0); g.set_nweight(0, 3, 0.25, 0.25)
    g.set_tweight(0, 10.0, 0.0); g.set_tweight(5, 0.0, 10.0)
    vg = g.get_graph()
    assert vg.get_edge(0, 1) == 3.0 and vg.get_edge(1, 0) == 2.5 and vg.get_edge(1, 4) == 0.0
    o = bk.BKGraph(6, 7)
    o.sum_edges([0, 0, 1, 2, 0], [1, 1, 2, 5, 3], [2.0, 1.0, 1.0, 4.0, 0.25], [2.0, 0.5, 1.0, 4.0, 0.25])
    o.add_tweights([0, 5], [10.0, 0.0], [0.0, 10.0])
    assert vg.maxflow() == o.maxflow()
    np.testing.assert_array_equal(vg.labels().ravel().astype(np.uint8), o.labels())
    with pytest.raises(AttributeError):
        gc.graph_from_voxels(np.zeros((2, 2)), np.zeros((2, 2)), boundary_term=lambda a: None)


def test_boundary_image_of_another_shape_like_the_reference_tests():
    """reference tests/graphcut_/energy_voxel.py:152-179 (__test_all_on_image): 4x4 markers with a 3x3 image, all eight
    terms must run; the reference numbers the edges by the image shape (energy_voxel.py:650-664).  Labels are compared
    with the oracle built the same way (lattice of the image shape over ids 0..8, ids 9..15 isolated)."""
    gc = _gc()
    from oracle import bk
    fgm = np.zeros((4, 4), bool); fgm[3, 3] = True
    bgm = np.zeros((4, 4), bool); bgm[0, 0] = True
    for image in (np.asarray([[-1, 1, -4], [2, -7, 3], [-2.3, 3, -7]], dtype=float), np.zeros((3, 3))):
        for term in TERMS:
            g = gc.graph_from_voxels(fgm, bgm, boundary_term=_term_fn(term), boundary_term_args=_term_args(term, image, 1.0, False))
            g.maxflow()
            res = np.array([0 if g.termtype.SINK == g.what_segment(i) else 1 for i in range(16)])
            assert res.shape == (16,)
            with np.errstate(all="ignore"):
                w = energy_numpy.boundary_weights(term, image, 1.0)
            if not np.isnan(np.concatenate([x.ravel() for x in w])).any():
                o = bk.BKGraph(16, 24)
                o.sum_lattice((3, 3), w)
                o.add_tweights([15], [65535.0], [0.0]); o.add_tweights([0], [0.0], [65535.0])
                o.maxflow()
                np.testing.assert_array_equal(res, o.labels())
    with pytest.raises(ValueError):
        gc.graph_from_voxels(np.zeros((2, 2)), np.zeros((2, 2)), boundary_term=_term_fn("difference_linear"),
                             boundary_term_args=(np.zeros((3, 3)), False))


@pytest.mark.parametrize("term,sigma", [("difference_exponential", 15.0), ("maximum_exponential", 400.0), ("difference_power", 1.7), ("maximum_power", 0.6)])
@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.int16, np.float32])
def test_integer_valued_images_get_bit_identical_exp_and_pow_weights(term, sigma, dtype):
    """Integer-valued images (CT / MR): the exponential and power terms are evaluated by a table the host fills with the
    reference's own NumPy operations (mgc_set_boundary_lut; reference energy_voxel.py:226-236, 290-300, 444-452, 506-513), so
    the n-link weights equal the reference's BIT FOR BIT (the device's exp / pow alone are up to 2 ulp away), with and without
    spacing; the labels equal BK's, or differ only where the reference's own residual graph is ambiguous (exact ties)."""
    rng = np.random.default_rng(5)
    shape = (12, 20, 17)
    hi = 200 if dtype in (np.uint8,) else 3000
    lo = -1500 if dtype in (np.int16, np.float32) else 0
    img = rng.integers(lo, hi, shape).astype(dtype)
    fg = np.zeros(shape, bool); fg[5:7, 9:11, 8:10] = True
    bg = np.zeros(shape, bool); bg[0] = bg[-1] = True; bg[:, 0] = bg[:, -1] = True; bg[:, :, 0] = bg[:, :, -1] = True
    for spacing in (False, (1.0, 0.7, 2.5)):
        g = _run(fg, bg, term, img, sigma, spacing=spacing)
        want = energy_numpy.boundary_weights(term, img, sigma, spacing)
        for axis in range(3):
            got = g.nweights(axis)
            assert got.tobytes() == np.ascontiguousarray(want[axis], dtype=np.float64).tobytes(), (term, dtype, spacing, axis)
        flow = g.maxflow()
        ref = pipeline.graphcut_voxel(fg, bg, term=term, image=img, sigma=sigma, spacing=spacing)
        labels = g.labels()
        if (labels != ref.labels).any():
            # identical capacities, but random whole-number images repeat weights everywhere: exact ties, broken by the order
            # in which a solver adds its flows.  A differing voxel must lie in the reference's own ambiguity set, and the cuts
            # must have the same capacity as exact rationals (oracle/cutcheck.py)
            i, j, ww = cutcheck.lattice_edges(shape, want)
            tr = np.where(fg, 65535.0, 0.0) - np.where(bg, 65535.0, 0.0)
            cutcheck.assert_labels_equivalent(labels, ref, exact=(i, j, ww, ww, tr))
        assert flow == pytest.approx(ref.flow, rel=1e-12)


def test_ties_volume_capacities_are_the_reference_doubles():
    """the tie-heavy synthetic (intensities 0..3 held in float32, sigma 1): with table-evaluated weights every capacity is the
    reference's own double (rounds 1-3: <= 2 ulp off); what is left between the two label volumes is summation order on exact
    ties, inside the ambiguity set"""
    from medpy_amd import synthetic
    s = synthetic.ties((40, 40, 40))
    g = _run(s["fg"], s["bg"], s["term"], s["image"], s["sigma"])
    want = energy_numpy.boundary_weights(s["term"], s["image"], s["sigma"])
    for axis in range(3):
        assert g.nweights(axis).tobytes() == np.ascontiguousarray(want[axis], dtype=np.float64).tobytes()
    g.maxflow()
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term=s["term"], image=s["image"], sigma=s["sigma"])
    labels = g.labels()
    if (labels != ref.labels).any():
        i, j, ww = cutcheck.lattice_edges(s["image"].shape, want)
        tr = np.where(s["fg"], 65535.0, 0.0) - np.where(s["bg"], 65535.0, 0.0)
        cutcheck.assert_labels_equivalent(labels, ref, exact=(i, j, ww, ww, tr))


@pytest.mark.parametrize("conn", [6, 26])
def test_prepush_gives_the_same_cut(conn):
    """k_build's pre-push (graphs with a regional term: source -> u -> v -> sink paths settled inside a tile while its weights are
    at hand) only changes the preflow a solve starts from: labels and flow with it, without it, and from the BK oracle agree; the
    invariant check (conservation against the capacities AS BUILT) holds on the pre-pushed, solved graph."""
    from medpy_amd import _lib, synthetic
    from medpy_amd.graphcut.graph import VoxelGraph
    shape = (56, 48, 64)
    s, r = synthetic.sphere(shape), synthetic.regional(shape)
    out = {}
    for pre in (1, 0):
        g = VoxelGraph(shape, connectivity=conn if conn != 6 else None)
        g._set_boundary(s["term"], s["image"], s["sigma"], False)
        g._set_regional(r["prob"], r["alpha"])
        g._set_markers(s["fg"], s["bg"])
        g.set_param("prepush", pre)
        g._build()
        flow = g.maxflow()
        _lib.assert_valid(g.validate())
        out[pre] = (g.labels(), flow)
        g.close()
    np.testing.assert_array_equal(out[1][0], out[0][0])
    assert out[1][1] == pytest.approx(out[0][1], rel=1e-12)
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term=s["term"], image=s["image"], sigma=s["sigma"], prob=r["prob"], alpha=r["alpha"],
                                  connectivity=conn if conn != 6 else None)
    np.testing.assert_array_equal(out[1][0], ref.labels)
    assert out[1][1] == pytest.approx(ref.flow, rel=1e-9)
