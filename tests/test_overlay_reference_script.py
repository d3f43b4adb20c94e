"""the reference's OWN command-line script, unmodified, on top of this package.

``medpy_amd.overlay.install()`` routes ``medpy.graphcut`` (and, where no MedPy is installed, ``medpy.core`` / ``medpy.io``) to
this package; then /root/reference/bin/medpy_graphcut_voxel.py is executed as it stands (runpy) on the reference's
notebook fixture b0 and must write the segmentation the reference pipeline produced (tests/golden/reference_b0.npz).

The build container has /root/reference but no GPU, the GPU box the opposite, so the solver object behind the facade is
the host-simulator stand-in here (tests/hostsim/sim_backend.py: same tile operations, run as loops) -- everything above it
is the product: overlay, graph_from_voxels, the energy_voxel plug-in protocol, GCGraph, termtype, what_segment, io."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SCRIPT = "/root/reference/bin/medpy_graphcut_voxel.py"


@pytest.fixture
def overlay_with_sim_backend(monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
    import sim_backend
    from medpy_amd import overlay
    from medpy_amd.graphcut import graph
    saved = {k: v for k, v in sys.modules.items() if k == "medpy" or k.startswith("medpy.")}
    for k in saved:
        del sys.modules[k]
    monkeypatch.setattr(graph, "VoxelGraph", sim_backend.SimVoxelGraph)
    mode = overlay.install(force_shim=True)
    yield overlay, mode
    for k in [k for k in sys.modules if k == "medpy" or k.startswith("medpy.")]:
        del sys.modules[k]
    sys.modules.update(saved)


@pytest.mark.needs_reference
def test_reference_voxel_script_runs_unmodified(tmp_path, overlay_with_sim_backend):
    overlay, mode = overlay_with_sim_backend
    assert mode == "shim"
    from conftest import GOLDEN
    from medpy_amd import io
    z = np.load(os.path.join(GOLDEN, "reference_b0.npz"))
    img = z["image"].astype(np.dtype(str(z["image_dtype"])))
    markers = z["markers"]
    # a 256x256 crop keeps the simulator run and the script's one-call-per-voxel read-out loop short
    img, markers = np.ascontiguousarray(img[384:640, 384:640]), np.ascontiguousarray(markers[384:640, 384:640])
    hdr = io.Header((1.0, 1.0))
    io.save(img, str(tmp_path / "b0.nii.gz"), hdr, True)
    io.save(markers, str(tmp_path / "markers.nii.gz"), hdr, True)
    out = str(tmp_path / "seg.nii.gz")
    overlay.run(REF_SCRIPT, ["10", str(tmp_path / "b0.nii.gz"), str(tmp_path / "markers.nii.gz"), out, "--boundary", "diff_exp", "-f"])
    seg, _ = io.load(out)
    from oracle import pipeline
    ref = pipeline.graphcut_voxel(markers == 1, markers == 2, term="difference_exponential", image=img, sigma=10.0)
    assert seg.shape == img.shape and seg.dtype == np.uint8
    # integer-valued image: exact ties between minimum cuts -- any differing voxel must be ambiguous in the reference's own
    # residual graph, at equal cut capacity (oracle/cutcheck.py); this crop differs in one
    from oracle import cutcheck, energy_numpy
    i, j, ww = cutcheck.lattice_edges(img.shape, energy_numpy.boundary_weights("difference_exponential", img, 10.0))
    tr = np.where(markers == 1, 65535.0, 0.0) - np.where(markers == 2, 65535.0, 0.0)
    cutcheck.assert_labels_equivalent(seg.astype(bool), ref, exact=(i, j, ww, ww, tr))
    # and the script's refusal to overwrite (save.py:75-78 through medpy_amd.io) surfaces as the reference's exit path
    with pytest.raises((SystemExit, Exception)):
        overlay.run(REF_SCRIPT, ["10", str(tmp_path / "b0.nii.gz"), str(tmp_path / "markers.nii.gz"), out, "--boundary", "diff_exp"])


def test_overlay_module_surface():
    """what the reference scripts import (bin/medpy_graphcut_voxel.py:20-37) exists after install(), without a GPU"""
    from medpy_amd import overlay
    saved = {k: v for k, v in sys.modules.items() if k == "medpy" or k.startswith("medpy.")}
    for k in saved:
        del sys.modules[k]
    try:
        assert overlay.install(force_shim=True) == "shim"
        from medpy import graphcut
        from medpy.core import ArgumentError, Logger
        from medpy.graphcut.wrapper import split_marker
        from medpy.io import header, load, save
        import medpy_amd.graphcut
        assert graphcut is medpy_amd.graphcut and graphcut.energy_voxel.boundary_difference_exponential
        assert issubclass(ArgumentError, Exception) and Logger.getInstance() is Logger.getInstance()
        assert callable(split_marker) and callable(load) and callable(save) and callable(header.get_pixel_spacing)
        from medpy.graphcut.maxflow import GraphDouble
        assert GraphDouble is medpy_amd.graphcut.GraphDouble
    finally:
        for k in [k for k in sys.modules if k == "medpy" or k.startswith("medpy.")]:
            del sys.modules[k]
        sys.modules.update(saved)


SHIPPED_SCRIPT = os.path.join(ROOT, "oracle", "_ref", "bin", "medpy_graphcut_voxel.py")  # written by __graft_entry__.build()


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(SHIPPED_SCRIPT), reason="oracle/_ref/bin/medpy_graphcut_voxel.py did not travel (run __graft_entry__.build() where /root/reference exists)")
@pytest.mark.parametrize("flag,term", [("diff_exp", "difference_exponential"), ("diff_div", "difference_division"),
                                       ("diff_pow", "notebook_gradient_diff_pow"), ("max_div", "notebook_b0_max_div")])
def test_reference_voxel_script_unmodified_on_the_hip_path(tmp_path, flag, term):
    """The reference's bin/medpy_graphcut_voxel.py, byte for byte (shipped to the GPU box as a build product), run by
    medpy_amd.overlay over libmedpyhip.so on the reference's notebook image b0 (1024 x 1024): its argument parsing, its
    load / split_marker / graph_from_voxels / maxflow / one-call-per-voxel what_segment loop / save
    (bin/medpy_graphcut_voxel.py:139-187) -- and the file it writes must hold the segmentation the reference pipeline wrote
    for the same command (tests/golden/reference_b0.npz), up to the voxels the reference's own residual graph leaves open."""
    from conftest import GOLDEN
    from medpy_amd import io, overlay
    from oracle import cutcheck, energy_numpy, pipeline
    saved = {k: v for k, v in sys.modules.items() if k == "medpy" or k.startswith("medpy.")}
    for k in saved:
        del sys.modules[k]
    try:
        z = np.load(os.path.join(GOLDEN, "reference_b0.npz"))
        img = z["image"].astype(np.dtype(str(z["image_dtype"])))
        markers = z["markers"]
        sigma = float(z[term + "/sigma"])
        key = term
        if term.startswith("notebook_"):
            # the notebook's two literal command lines (reference notebooks/scripts/medpy_graphcut_voxel.py.ipynb):
            #   medpy_graphcut_voxel.py 10 gradient.nii.gz b0markers.nii.gz out --boundary diff_pow -f    (gradient = medpy_gradient.py b0)
            #   medpy_graphcut_voxel.py 1 b0.nii.gz b0markers.nii.gz out --boundary=max_div -f
            term = str(z[key + "/term"])
            if "gradient" in key:  # bin/medpy_gradient.py:77-83
                from scipy.ndimage import generic_gradient_magnitude, prewitt
                grad = np.zeros(img.shape, dtype=np.float32)
                generic_gradient_magnitude(img, prewitt, output=grad)
                img = grad
        hdr = io.Header((1.0, 1.0))
        io.save(img, str(tmp_path / "b0.nii.gz"), hdr, True)
        io.save(markers, str(tmp_path / "b0markers.nii.gz"), hdr, True)
        out = str(tmp_path / "seg.nii.gz")
        argv = [("%g" % sigma), str(tmp_path / "b0.nii.gz"), str(tmp_path / "b0markers.nii.gz"), out] + (["--boundary=" + flag] if key == "notebook_b0_max_div" else ["--boundary", flag]) + ["-f"]
        overlay.run(SHIPPED_SCRIPT, argv)
        import medpy.graphcut
        import medpy_amd.graphcut
        assert medpy.graphcut is medpy_amd.graphcut  # the script's `from medpy import graphcut` got this package
        seg, _ = io.load(out)
        ref = np.unpackbits(z[key + "/labels"])[: img.size].reshape(img.shape)
        assert seg.shape == img.shape and seg.dtype == np.uint8
        nbad = int((seg != ref).sum())
        print(term, "voxels differing from the reference's output:", nbad)
        if nbad:
            fg, bg = markers == 1, markers == 2
            cut = pipeline.graphcut_voxel(fg, bg, term=term, image=img, sigma=sigma)
            np.testing.assert_array_equal(cut.labels, ref.astype(bool))
            i, j, ww = cutcheck.lattice_edges(img.shape, energy_numpy.boundary_weights(term, img, sigma))
            tr = np.where(fg, 65535.0, 0.0) - np.where(bg, 65535.0, 0.0)
            cutcheck.assert_labels_equivalent(seg.astype(bool), cut, exact=(i, j, ww, ww, tr))
    finally:
        for k in [k for k in sys.modules if k == "medpy" or k.startswith("medpy.")]:
            del sys.modules[k]
        sys.modules.update(saved)
