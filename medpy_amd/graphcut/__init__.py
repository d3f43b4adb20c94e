"""medpy_amd.graphcut -- ``medpy.graphcut`` on MI355X.

Same public names as reference medpy/graphcut/__init__.py: ``graph_from_voxels`` / ``energy_voxel`` (voxel lattices:
the tile solver), ``graph_from_labels`` / ``energy_label`` (region graphs: the sparse-graph solver), ``GCGraph``,
``split_marker``, ``graphcut_stawiaski``; ``GraphDouble`` / ``GraphFloat`` / ``GraphInt`` are the solver classes for
arbitrary graphs (reference lib/maxflow/src/wrapper.cpp:27-134).
"""
from . import energy_label, energy_voxel
from .generate import graph_from_labels, graph_from_voxels
from .graph import GCGraph, Graph, GraphFloat, GraphInt, SparseGraph, VoxelGraph, termtype
from .write import graph_to_dimacs
from .wrapper import graphcut_stawiaski, split_marker

GraphDouble = SparseGraph

__all__ = ["graph_from_voxels", "graph_from_labels", "energy_voxel", "energy_label", "GCGraph", "VoxelGraph", "SparseGraph",
           "GraphDouble", "GraphFloat", "GraphInt", "Graph", "graph_to_dimacs", "termtype", "split_marker", "graphcut_stawiaski"]
