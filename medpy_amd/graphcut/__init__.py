"""medpy_amd.graphcut -- the voxel half of ``medpy.graphcut`` on MI355X.

Same public names as reference medpy/graphcut/__init__.py for the voxel path:
``graph_from_voxels``, ``energy_voxel``, ``GCGraph``, ``split_marker``; ``GraphDouble`` is the
class of the returned solver object.
"""
from . import energy_voxel
from .generate import graph_from_voxels
from .graph import GCGraph, VoxelGraph, termtype
from .wrapper import split_marker

GraphDouble = VoxelGraph

__all__ = ["graph_from_voxels", "energy_voxel", "GCGraph", "VoxelGraph", "GraphDouble", "termtype", "split_marker"]
