"""-m gpu: the Z-slab (multi-GPU) code path on ONE MI355X: N slab handles time-multiplexed on the
device, borders exchanged through the loopback transport (host buffers and HBM buffers).  Labels
must equal the single-handle labels and the oracle's, bit for bit."""
import numpy as np
import pytest

from oracle import pipeline

pytestmark = pytest.mark.gpu


def _single(s):
    from medpy_amd import graphcut
    g = graphcut.graph_from_voxels(s["fg"], s["bg"], boundary_term=graphcut.energy_voxel.boundary_difference_exponential,
                                   boundary_term_args=(s["image"], s["sigma"], False))
    flow = g.maxflow()
    return g.labels(), flow


@pytest.mark.parametrize("gen,shape,nslabs,devbuf", [("sphere", (64, 48, 40), 2, False), ("sphere", (64, 48, 40), 4, True),
                                                     ("hard", (48, 48, 48), 3, False), ("sphere", (37, 40, 24), 2, True),
                                                     ("sphere", (128, 64, 64), 8, True)])
def test_slabs_equal_single_and_oracle(gen, shape, nslabs, devbuf):
    from medpy_amd import synthetic
    from medpy_amd.slab import graphcut_voxel_slabs
    s = getattr(synthetic, gen)(shape)
    labels, flow, st = graphcut_voxel_slabs(s["image"], s["fg"], s["bg"], s["term"], s["sigma"], nslabs=nslabs, device_buffers=devbuf)
    assert st["converged"] == 1
    single, sflow = _single(s)
    np.testing.assert_array_equal(labels, single)
    ref = pipeline.graphcut_voxel(s["fg"], s["bg"], term=s["term"], image=s["image"], sigma=s["sigma"])
    np.testing.assert_array_equal(labels, ref.labels)
    assert flow == pytest.approx(ref.flow, rel=1e-9)
    assert flow == pytest.approx(sflow, rel=1e-12)


def test_slab_handle_refuses_single_gpu_entry_points():
    from medpy_amd import _lib
    from medpy_amd.slab import HipSlab
    s = HipSlab((32, 16, 16), 0, 2)
    assert (s.own0, s.own1, s.plane0, s.plane1) == (0, 16, 0, 24) and s.has_hi and not s.has_lo
    s.set_boundary("difference_linear", np.zeros(s.local_shape, np.float32), None)
    s.build()
    import ctypes as C
    with pytest.raises(_lib.MedpyHipError):
        s._call("mgc_maxflow", C.byref(C.c_double()))
    s.close()
