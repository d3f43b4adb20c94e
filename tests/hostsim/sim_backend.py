"""TEST INFRASTRUCTURE ONLY: a stand-in for medpy_amd.graphcut.graph.VoxelGraph that solves on the HOST SIMULATOR
(tests/hostsim: the solver's tile operations executed as plain loops) with the oracle's NumPy energies.  It exists so
that the drop-in overlay -- the reference's own command-line script running on top of medpy_amd.graphcut -- can be
executed in the build container, which has /root/reference but no GPU (the GPU box has a GPU but no /root/reference).
The product never imports this: without the HIP library VoxelGraph raises."""
import numpy as np

import sim
from medpy_amd.graphcut.graph import termtype as _termtype
from oracle import energy_numpy


class SimVoxelGraph(object):
    termtype = _termtype

    def __init__(self, shape, device=0, connectivity=None):
        assert connectivity in (None, 2 * len(shape)), "the stand-in covers the reference's own neighbourhood"
        self.shape = tuple(int(s) for s in shape)
        self._boundary = None
        self._fg = self._bg = None
        self._labels = None

    def _set_boundary(self, term, image, sigma, spacing):
        self._boundary = (term, np.asarray(image), sigma, spacing)

    def _set_markers(self, fg, bg):
        self._fg = None if fg is None else np.asarray(fg).reshape(self.shape).astype(bool)
        self._bg = None if bg is None else np.asarray(bg).reshape(self.shape).astype(bool)

    def _build(self):
        pass

    def maxflow(self):
        term, image, sigma, spacing = self._boundary
        w = energy_numpy.boundary_weights(term, image, sigma, spacing)
        shp = (1,) * (3 - len(self.shape)) + self.shape
        w3 = [np.zeros(0)] * (3 - len(self.shape)) + [np.asarray(x) for x in w]
        tr = np.zeros(self.shape)
        if self._fg is not None:
            tr = np.where(self._fg, 65535.0, tr)
        if self._bg is not None:
            tr = np.where(self._bg & ~(self._fg if self._fg is not None else False), -65535.0, np.where(self._bg, 0.0, tr))
        lab, st = sim.solve(shp, w3, tr, wave_mode=1)
        assert st["converged"] == 1
        self._labels = lab.reshape(self.shape).astype(bool)
        return 0.0

    def labels(self):
        return self._labels

    def what_segment(self, i):
        return _termtype.SOURCE if self._labels.flat[i] else _termtype.SINK
