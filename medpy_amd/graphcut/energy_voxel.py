"""Voxel energy terms: same names, signatures and argument tuples as reference
medpy/graphcut/energy_voxel.py, usable as ``boundary_term`` / ``regional_term`` plug-ins of
``graph_from_voxels``.

The reference evaluates g(.) with NumPy and then inserts every edge with one Python call
(energy_voxel.py:660-664).  Here a term only *records* itself on the graph facade; the
weights are computed by the HIP kernel ``k_build`` (medpy_amd/csrc/mgc_kernels.hip) for all
six neighbours of every voxel while the residual lattice is laid out in HBM.
"""
import numpy

from .graph import GCGraph

__all__ = [
    "regional_probability_map",
    "boundary_maximum_linear", "boundary_difference_linear",
    "boundary_maximum_exponential", "boundary_difference_exponential",
    "boundary_maximum_division", "boundary_difference_division",
    "boundary_maximum_power", "boundary_difference_power",
]


def _need_facade(graph):
    if not isinstance(graph, GCGraph):
        raise TypeError("medpy_amd energy terms work on medpy_amd.graphcut.GCGraph objects "
                        "(as created by medpy_amd.graphcut.graph_from_voxels), got {}".format(type(graph)))


def regional_probability_map(graph, xxx_todo_changeme):
    """Regional term from a foreground probability map; reference energy_voxel.py:33-65.

    t-links (p*alpha, (1-p)*alpha) per voxel, evaluated in the dtype of the map as NumPy does.
    """
    (probability_map, alpha) = xxx_todo_changeme
    _need_facade(graph)
    graph.record_regional(numpy.asarray(probability_map), alpha)


def boundary_maximum_linear(graph, xxx_todo_changeme1):
    """w = 1 - max(|Ip|,|Iq|) / max|I|; reference energy_voxel.py:68-114."""
    (gradient_image, spacing) = xxx_todo_changeme1
    _need_facade(graph)
    graph.record_boundary("maximum_linear", gradient_image, None, spacing)


def boundary_difference_linear(graph, xxx_todo_changeme2):
    """w = 1 - |Ip-Iq| / |max I - min I|; reference energy_voxel.py:117-191."""
    (original_image, spacing) = xxx_todo_changeme2
    _need_facade(graph)
    graph.record_boundary("difference_linear", original_image, None, spacing)


def boundary_maximum_exponential(graph, xxx_todo_changeme3):
    """w = exp(-max(|Ip|,|Iq|)^2 / sigma^2); reference energy_voxel.py:194-238."""
    (gradient_image, sigma, spacing) = xxx_todo_changeme3
    _need_facade(graph)
    graph.record_boundary("maximum_exponential", gradient_image, sigma, spacing)


def boundary_difference_exponential(graph, xxx_todo_changeme4):
    """w = exp(-|Ip-Iq|^2 / sigma^2); reference energy_voxel.py:241-302."""
    (original_image, sigma, spacing) = xxx_todo_changeme4
    _need_facade(graph)
    graph.record_boundary("difference_exponential", original_image, sigma, spacing)


def boundary_maximum_division(graph, xxx_todo_changeme5):
    """w = 1 / (1 + |Ip-Iq| / sigma): the reference routes this through the *difference*
    skeleton (energy_voxel.py:347); kept for parity.  Reference energy_voxel.py:305-347."""
    (gradient_image, sigma, spacing) = xxx_todo_changeme5
    _need_facade(graph)
    graph.record_boundary("maximum_division", gradient_image, sigma, spacing)


def boundary_difference_division(graph, xxx_todo_changeme6):
    """w = 1 / (1 + |Ip-Iq| / sigma); reference energy_voxel.py:350-409."""
    (original_image, sigma, spacing) = xxx_todo_changeme6
    _need_facade(graph)
    graph.record_boundary("difference_division", original_image, sigma, spacing)


def boundary_maximum_power(graph, xxx_todo_changeme7):
    """w = (1 / (1 + max(|Ip|,|Iq|)))^sigma; reference energy_voxel.py:412-452."""
    (gradient_image, sigma, spacing) = xxx_todo_changeme7
    _need_facade(graph)
    graph.record_boundary("maximum_power", gradient_image, sigma, spacing)


def boundary_difference_power(graph, xxx_todo_changeme8):
    """w = (1 / (1 + |Ip-Iq|))^sigma; reference energy_voxel.py:455-516."""
    (original_image, sigma, spacing) = xxx_todo_changeme8
    _need_facade(graph)
    graph.record_boundary("difference_power", original_image, sigma, spacing)
