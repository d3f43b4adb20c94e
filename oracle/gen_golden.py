"""Generate tests/golden/*.npz by running the REFERENCE ITSELF in this container.

    python -m oracle.gen_golden            (needs /root/reference; see oracle/overlay.py)

Every fixture stores inputs plus what the reference's own code produced for them:
``medpy.graphcut.graph_from_voxels`` (generate.py:33-174) with the reference energy terms
(energy_voxel.py) feeding the unmodified BK solver, then ``maxflow()`` and the
``what_segment`` read-out of bin/medpy_graphcut_voxel.py:172-182.  The n-link weights are
captured with a recording subclass of the reference ``GCGraph`` (the pattern of reference
tests/graphcut_/energy_label.py:196-215) that also forwards to the real graph.

The committed fixtures are what ``-m "not gpu"`` tests pin the oracle against and what the
``-m gpu`` tests pin the HIP path against on the GPU box, where /root/reference is absent.
TEST INFRASTRUCTURE ONLY.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, os.path.dirname(HERE))

from oracle.overlay import import_reference_graphcut  # noqa: E402

TERMS = ["difference_linear", "difference_exponential", "difference_division", "difference_power",
         "maximum_linear", "maximum_exponential", "maximum_division", "maximum_power"]


def run_reference(gc, fg, bg, term=None, image=None, sigma=None, spacing=False, prob=None, alpha=None):
    """Returns dict(labels, flow, trcap, edges_i, edges_j, edges_w) from the reference pipeline."""
    import medpy.graphcut.generate as gen
    import medpy.graphcut.graph as graphmod

    rec = {}

    class Recorder(graphmod.GCGraph):
        def set_nweight(self, a, b, w, wr):
            graphmod.GCGraph.set_nweight(self, a, b, w, wr)
            rec[(int(a), int(b))] = rec.get((int(a), int(b)), 0.0) + float(w)

    orig = gen.GCGraph
    gen.GCGraph = Recorder
    try:
        kw = {}
        if term is not None:
            fn = getattr(gc.energy_voxel, "boundary_" + term)
            kw["boundary_term"] = fn
            kw["boundary_term_args"] = (image, spacing) if term.endswith("linear") else (image, sigma, spacing)
        if prob is not None:
            kw["regional_term"] = gc.energy_voxel.regional_probability_map
            kw["regional_term_args"] = (prob, alpha)
        with np.errstate(all="ignore"):
            g = gc.graph_from_voxels(fg, bg, **kw)
    finally:
        gen.GCGraph = orig
    n = g.get_node_num()
    trcap = np.array([g.get_trcap(i) for i in range(n)])
    flow = g.maxflow()
    labels = np.array([0 if g.termtype.SINK == g.what_segment(i) else 1 for i in range(n)], dtype=np.uint8)
    keys = sorted(rec)
    return {
        "labels": labels.reshape(np.asarray(fg).shape),
        "flow": np.float64(flow),
        "trcap": trcap,
        "edges_i": np.array([k[0] for k in keys], dtype=np.int64),
        "edges_j": np.array([k[1] for k in keys], dtype=np.int64),
        "edges_w": np.array([rec[k] for k in keys], dtype=np.float64),
    }


def pack(prefix, d, store):
    for k, v in d.items():
        store["%s/%s" % (prefix, k)] = v


def main():
    gc = import_reference_graphcut()
    os.makedirs(OUT, exist_ok=True)

    # ---- 1. the reference's own known-answer inputs (tests/graphcut_/cut.py:32-50,
    #         tests/graphcut_/energy_voxel.py:55-66,110-149): inputs restated, outputs from the reference
    store = {}
    vol = np.asarray([[[1, 0, 1, 2, 3], [1, 0, 1, 4, 3], [0, 1, 1, 6, 4]]] * 2)
    vfg = np.zeros(vol.shape, int); vfg[:, 2, 0] = 1
    vbg = np.zeros(vol.shape, int); vbg[:, 0, 4] = 1
    store["cut/image"], store["cut/fg"], store["cut/bg"] = vol, vfg, vbg
    pack("cut", run_reference(gc, vfg, vbg, "difference_linear", vol), store)

    img = np.asarray([[0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 1, 1], [0, 0, 1, 1]], dtype=float)
    grad = np.asarray([[0, 0, 0, 0], [0, 1, 1, 1], [0, 1, 0, 0], [0, 1, 0, 0]], dtype=float)
    fg = np.zeros((4, 4), int); fg[3, 3] = 1
    bg = np.zeros((4, 4), int); bg[0, 0] = 1
    store["e2d/image"], store["e2d/gradient"], store["e2d/fg"], store["e2d/bg"] = img, grad, fg, bg
    sig = {"exponential": 1.0, "division": 0.5, "power": 2.0, "linear": None}
    for t in TERMS:
        im = img if t.startswith("difference") else grad
        pack("e2d/" + t, run_reference(gc, fg, bg, t, im, sig[t.split("_")[1]]), store)
    pack("e2d/regional", run_reference(gc, fg, bg, prob=img / 2.0, alpha=1.0), store)

    simg = np.zeros((5, 5)); simg[1:, 2] = 2
    sfg = np.zeros((5, 5), bool); sfg[4, 2] = 1
    sbg = np.zeros((5, 5), bool); sbg[0, 0] = sbg[0, 4] = 1
    store["spacing/image"], store["spacing/fg"], store["spacing/bg"] = simg, sfg, sbg
    pack("spacing", run_reference(gc, sfg, sbg, "difference_division", simg, 1.0, (1.0, 5.0)), store)
    np.savez_compressed(os.path.join(OUT, "reference_kat.npz"), **store)

    # ---- 2. seeded small volumes through all eight terms, several dtypes, spacing on/off, + regional
    store = {}
    rng = np.random.default_rng(1234)
    cases = []
    for idx, (shape, dtype) in enumerate([((5, 6, 7), np.float32), ((4, 9, 3), np.float64), ((6, 5, 8), np.uint16),
                                          ((7, 1, 6), np.float32), ((3, 4, 5, 2), np.float64), ((9, 11), np.int16)]):
        if np.issubdtype(dtype, np.floating):
            image = (rng.normal(0, 20, shape) + 50 * (rng.random(shape) < 0.4)).astype(dtype)
        elif dtype == np.uint16:
            image = rng.integers(0, 6, shape).astype(dtype)
        else:
            image = rng.integers(-5, 6, shape).astype(dtype)
        u = rng.random(shape)
        fgm, bgm = u < 0.08, (u > 0.9)
        spacing = False if idx % 2 == 0 else tuple(float(x) for x in rng.uniform(0.5, 3.0, len(shape)))
        sigma = float(rng.uniform(0.5, 20.0))
        name = "c%d" % idx
        store[name + "/image"], store[name + "/fg"], store[name + "/bg"] = image, fgm, bgm
        store[name + "/sigma"] = np.float64(sigma)
        store[name + "/spacing"] = np.asarray(spacing if spacing else [], dtype=np.float64)
        for t in TERMS:
            pack("%s/%s" % (name, t), run_reference(gc, fgm, bgm, t, image, sigma, spacing), store)
        cases.append(name)
    # regional (+ boundary) with float32 and float64 maps; fg&bg overlap voxel to pin t-link merging (graph.h:416-425)
    for idx, dtype in enumerate([np.float32, np.float64]):
        shape = (5, 6, 4)
        image = rng.normal(0, 10, shape).astype(np.float32)
        prob = np.clip(rng.normal(0.5, 0.3, shape), 0, 1).astype(dtype)
        fgm = rng.random(shape) < 0.1
        bgm = rng.random(shape) < 0.1
        fgm[0, 0, 0] = bgm[0, 0, 0] = True
        name = "r%d" % idx
        store[name + "/image"], store[name + "/fg"], store[name + "/bg"], store[name + "/prob"] = image, fgm, bgm, prob
        store[name + "/sigma"], store[name + "/alpha"] = np.float64(7.5), np.float64(0.5)
        pack(name + "/cut", run_reference(gc, fgm, bgm, "difference_exponential", image, 7.5, False, prob, 0.5), store)
        pack(name + "/regional_only", run_reference(gc, fgm, bgm, prob=prob, alpha=0.5), store)
    np.savez_compressed(os.path.join(OUT, "reference_small.npz"), **store)

    # ---- 3. synthetic lattices of SURVEY 8(d) at small sizes: labels + flow from the reference pipeline
    from medpy_amd import synthetic
    store = {}
    for gen_name, shape in [("sphere", (16, 16, 16)), ("sphere", (12, 20, 17)), ("hard", (16, 16, 16)), ("ties", (14, 14, 14))]:
        s = getattr(synthetic, gen_name)(shape)
        name = "%s_%s" % (gen_name, "x".join(map(str, shape)))
        r = run_reference(gc, s["fg"], s["bg"], s["term"], s["image"], s["sigma"])
        store[name + "/labels"], store[name + "/flow"] = np.packbits(r["labels"]), r["flow"]
        store[name + "/shape"] = np.asarray(shape)
    np.savez_compressed(os.path.join(OUT, "reference_synthetic.npz"), **store)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


def main_b0():
    """Real 2-D data: the reference's notebook fixture (notebooks/scripts/resources/b0.nii.gz + b0markers.nii.gz,
    1024x1024 uint16) through the reference pipeline; loaded the way medpy.io.load would hand it over ((x, y) view)."""
    from medpy_amd import io
    gc = import_reference_graphcut()
    res = os.path.join(os.environ.get("MEDPY_REFERENCE", "/root/reference"), "notebooks", "scripts", "resources")
    img, _ = io.load(os.path.join(res, "b0.nii.gz"))
    markers, _ = io.load(os.path.join(res, "b0markers.nii.gz"))
    fg, bg = (markers == 1), (markers == 2)
    store = {"image": np.ascontiguousarray(img).astype(np.uint8) if img.max() < 256 else np.ascontiguousarray(img),
             "image_dtype": np.asarray(str(img.dtype)), "markers": np.ascontiguousarray(markers).astype(np.uint8)}
    for term, sigma in (("difference_exponential", 10.0), ("difference_division", 10.0)):
        r = run_reference(gc, fg, bg, term, img, sigma)
        store[term + "/labels"] = np.packbits(r["labels"])
        store[term + "/flow"] = r["flow"]
        store[term + "/sigma"] = np.float64(sigma)
        print(term, "flow", r["flow"], "fg fraction", r["labels"].mean())
    # the notebook's two LITERAL invocations (notebooks/scripts/medpy_graphcut_voxel.py.ipynb):
    #   medpy_gradient.py b0 gradient ; medpy_graphcut_voxel.py 10 gradient b0markers out --boundary diff_pow
    #   medpy_graphcut_voxel.py 1 b0 b0markers out --boundary=max_div
    # The gradient image is what bin/medpy_gradient.py:77-83 computes (float32 Prewitt gradient magnitude); the test recomputes
    # it with the same SciPy call, only labels and flow are stored.
    from scipy.ndimage import generic_gradient_magnitude, prewitt
    grad = np.zeros(img.shape, dtype=np.float32)
    generic_gradient_magnitude(img, prewitt, output=grad)
    for key, term, image, sigma in (("notebook_gradient_diff_pow", "difference_power", grad, 10.0), ("notebook_b0_max_div", "maximum_division", img, 1.0)):
        r = run_reference(gc, fg, bg, term, image, sigma)
        store[key + "/labels"] = np.packbits(r["labels"])
        store[key + "/flow"] = r["flow"]
        store[key + "/sigma"] = np.float64(sigma)
        store[key + "/term"] = np.asarray(term)
        print(key, "flow", r["flow"], "fg fraction", r["labels"].mean())
    np.savez_compressed(os.path.join(OUT, "reference_b0.npz"), **store)
    print("reference_b0.npz", os.path.getsize(os.path.join(OUT, "reference_b0.npz")))


def label_cases():
    """deterministic label images (block partitions with a ragged edge), images and markers for the region graph cut"""
    cases = []
    for name, shape, block, dtype in (("l2d_f32", (24, 30), 4, np.float32), ("l2d_f64", (20, 20), 5, np.float64),
                                      ("l3d_f32", (12, 14, 10), 3, np.float32), ("l3d_i16", (10, 12, 12), 4, np.int16)):
        rng = np.random.default_rng(len(cases) + 7)
        idx = np.indices(shape)
        coarse = tuple((idx[d] + (idx[(d + 1) % len(shape)] // 7)) // block for d in range(len(shape)))  # ragged borders
        dims = [int(c.max()) + 1 for c in coarse]
        flat = np.ravel_multi_index(coarse, dims)
        _, lab = np.unique(flat, return_inverse=True)
        lab = (lab.reshape(shape) + 1).astype(np.int32)
        centre = np.asarray(shape) / 2.0
        r = np.sqrt(sum((idx[d] - centre[d]) ** 2 for d in range(len(shape))))
        img = 100.0 * (r < 0.3 * min(shape)) + rng.normal(0, 10, shape)
        grad = np.abs(np.gradient(img)[0]) * (1 if len(cases) % 2 else -1)  # signed values exercise the abs()
        if np.issubdtype(dtype, np.integer):
            img, grad = np.round(img).astype(dtype), np.round(grad).astype(dtype)
        else:
            img, grad = img.astype(dtype), grad.astype(dtype)
        fg = r < 0.12 * min(shape)
        bg = np.zeros(shape, bool)
        bg[0], bg[-1] = True, True
        prob = np.clip(0.3 + 0.4 * (r < 0.3 * min(shape)) + rng.normal(0, 0.1, shape), 0, 1).astype(np.float32 if dtype == np.float32 else np.float64)
        cases.append(dict(name=name, labels=lab, image=img, gradient=grad, fg=fg, bg=bg, prob=prob))
    return cases


def main_labels():
    """The region graph cut: reference graph_from_labels (generate.py:177-338) + energy_label.py terms + BK."""
    gc = import_reference_graphcut()
    from medpy.graphcut import energy_label as el
    store = {}
    for c in label_cases():
        lab = c["labels"]
        n = int(lab.max())
        for tname, fn, args in (("stawiaski", el.boundary_stawiaski, c["gradient"]),
                                # positive directedness cannot be run: the reference calls addition_directed_ltd with four
                                # arguments although it takes five (energy_label.py:304, 347) -> TypeError
                                ("stawiaski_directed_neg", el.boundary_stawiaski_directed, (c["gradient"], -0.5)),
                                ("difference_of_means", el.boundary_difference_of_means, c["image"]),
                                ("stawiaski_atlas", el.boundary_stawiaski, c["gradient"])):
            kw = dict(boundary_term=fn, boundary_term_args=args)
            if tname.endswith("atlas"):
                kw.update(regional_term=el.regional_atlas, regional_term_args=(c["prob"], 0.5))
            g = gc.graph_from_labels(lab, c["fg"], c["bg"], **kw)
            edges = np.array([[g.get_edge(i, j) for j in range(n)] for i in range(n)])
            trcap = np.array([g.get_trcap(i) for i in range(n)])
            flow = g.maxflow()
            seg = np.array([0 if g.what_segment(i) == g.termtype.SINK else 1 for i in range(n)], dtype=np.uint8)
            key = "%s/%s" % (c["name"], tname)
            store[key + "/edges"], store[key + "/trcap"], store[key + "/flow"], store[key + "/segments"] = edges, trcap, flow, seg
            print(key, "regions", n, "flow", flow, "fg regions", int(seg.sum()))
        for k in ("labels", "image", "gradient", "fg", "bg", "prob"):
            store["%s/%s" % (c["name"], k)] = c[k]
    np.savez_compressed(os.path.join(OUT, "reference_labels.npz"), **store)
    print("reference_labels.npz", os.path.getsize(os.path.join(OUT, "reference_labels.npz")))


def main_large():
    """Large synthetic volumes (the sizes the metric is quoted on).  The reference's Python loop would issue
    4e8 set_nweight calls at 512^3, so these come from the compiled UNMODIFIED reference solver (oracle/_ref, built from
    /root/reference by oracle/Makefile) fed the NumPy restatement of the energies (pinned bitwise against the reference's
    own Python by tests/test_oracle_golden.py).  Stored: SHA-256 of the packed label volume, flow, foreground count.

        python oracle/gen_golden.py large [case ...]      (cases: see CASES; default all; 512^3 needs ~40 GB of RAM)
    """
    import hashlib
    import json
    import time
    from medpy_amd import synthetic
    from oracle import bk, pipeline
    assert bk.available("ref"), "needs the compiled reference (oracle/_ref): run where /root/reference exists"
    CASES = {
        "sphere_512_6": dict(gen="sphere", shape=(512, 512, 512), conn=6, regional=False),
        "sphere_256_6": dict(gen="sphere", shape=(256, 256, 256), conn=6, regional=False),
        "hard_256_6": dict(gen="hard", shape=(256, 256, 256), conn=6, regional=False),
        "sphere_128_26": dict(gen="sphere", shape=(128, 128, 128), conn=26, regional=False),
        "sphere_256_26": dict(gen="sphere", shape=(256, 256, 256), conn=26, regional=False),
        "config3_256_26_regional": dict(gen="sphere", shape=(256, 256, 256), conn=26, regional=True),
        # discriminating cases (round-2 review: the three 256^3 sphere cuts above are all the geometric ball): weak contrast, so
        # the cut depends on the neighbourhood and on the regional term
        "hard_128_26": dict(gen="hard", shape=(128, 128, 128), conn=26, regional=False),
        "hard_192_26": dict(gen="hard", shape=(192, 192, 192), conn=26, regional=False),
        "hard_192_6": dict(gen="hard", shape=(192, 192, 192), conn=6, regional=False),
        "config3_hard_192_26_alpha005": dict(gen="hard", shape=(192, 192, 192), conn=26, regional=True, alpha=0.05),
        "config3_hard_256_26_alpha005": dict(gen="hard", shape=(256, 256, 256), conn=26, regional=True, alpha=0.05),
    }
    path = os.path.join(OUT, "reference_large.json")
    store = json.load(open(path)) if os.path.exists(path) else {}
    for name in (sys.argv[2:] or list(CASES)):
        c = CASES[name]
        s = getattr(synthetic, c["gen"])(c["shape"])
        kw = {}
        if c["regional"]:
            r = synthetic.regional(c["shape"])
            kw = dict(prob=r["prob"], alpha=c.get("alpha", r["alpha"]))
        t0 = time.time()
        cut = pipeline.graphcut_voxel(s["fg"], s["bg"], term=s["term"], image=s["image"], sigma=s["sigma"], kind="ref",
                                      connectivity=c["conn"] if c["conn"] != 6 else None, **kw)
        lab = np.packbits(cut.labels.astype(np.uint8).ravel())
        store[name] = {"gen": c["gen"], "shape": list(c["shape"]), "connectivity": c["conn"], "regional": c["regional"], "alpha": kw.get("alpha"),
                       "sha256_packed_labels": hashlib.sha256(lab.tobytes()).hexdigest(), "flow": cut.flow,
                       "foreground_voxels": int(cut.labels.sum()), "oracle_seconds": round(time.time() - t0, 1)}
        print(name, store[name], flush=True)
        del cut, lab
        json.dump(store, open(path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "b0":
        main_b0()
    elif len(sys.argv) > 1 and sys.argv[1] == "large":
        main_large()
    elif len(sys.argv) > 1 and sys.argv[1] == "labels":
        main_labels()
    else:
        main()
