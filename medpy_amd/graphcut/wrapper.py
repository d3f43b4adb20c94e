"""Marker helpers used by bin/medpy_graphcut_voxel.py; reference medpy/graphcut/wrapper.py."""
import numpy


def split_marker(marker, fg_id=1, bg_id=2):
    """Split a marker image into fg / bg binary masks; reference wrapper.py:39-69."""
    img_marker = numpy.asarray(marker)
    img_fgmarker = numpy.zeros(img_marker.shape, numpy.bool_)
    img_fgmarker[img_marker == fg_id] = True
    img_bgmarker = numpy.zeros(img_marker.shape, numpy.bool_)
    img_bgmarker[img_marker == bg_id] = True
    return img_fgmarker, img_bgmarker
