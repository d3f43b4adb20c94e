"""medpy_amd -- MI355X (gfx950) implementation of MedPy's voxel graph-cut hot path.

Drop-in for ``medpy.graphcut.graph_from_voxels`` + ``medpy.graphcut.energy_voxel`` + the
``lib/maxflow`` solve, behind a C-ABI shared library of hand-written HIP kernels
(``medpy_amd/csrc``, ``include/medpy_hip.h``).  See DESIGN.md.
"""
__version__ = "0.1.0"
