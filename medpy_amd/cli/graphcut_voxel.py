"""Voxel graph-cut command line: the reference's ``bin/medpy_graphcut_voxel.py`` on MI355X.

Same positional arguments, options and flow as reference bin/medpy_graphcut_voxel.py:77-249
(``sigma badditional markers output [--boundary ...] [-s] [-f] [-v] [-d]``); the differences are the I/O layer
(``medpy_amd.io`` instead of SimpleITK: .npy / NIfTI-1) and the read-out (bulk ``labels()`` instead of one
``what_segment`` call per voxel, :177-181).  ``--connectivity`` is an extension (full neighbourhood).
"""
import argparse
import logging
import os
from argparse import RawTextHelpFormatter

import numpy

from .. import graphcut
from ..graphcut import split_marker
from ..io import get_pixel_spacing, load, save

__description__ = """
Perform a binary graph cut using Boykov's max-flow/min-cut definition on the voxels of an image, on an AMD MI355X.
Drop-in for medpy_graphcut_voxel.py: the markers image holds 1 for foreground and 2 for background seeds.
"""

BOUNDARY_TERMS = {
    "diff_linear": ("boundary_difference_linear", "linear difference of intensities"),
    "diff_exp": ("boundary_difference_exponential", "exponential difference of intensities"),
    "diff_div": ("boundary_difference_division", "divided difference of intensities"),
    "diff_pow": ("boundary_difference_power", "power based / raised difference of intensities"),
    "max_linear": ("boundary_maximum_linear", "linear maximum of intensities"),
    "max_exp": ("boundary_maximum_exponential", "exponential maximum of intensities"),
    "max_div": ("boundary_maximum_division", "divided maximum of intensities"),
    "max_pow": ("boundary_maximum_power", "power based / raised maximum of intensities"),
}


def main(argv=None):
    args = getArguments(getParser(), argv)
    logger = logging.getLogger("medpy_amd")
    logging.basicConfig(format="%(levelname)s: %(message)s")
    if args.debug:
        logger.setLevel(logging.DEBUG)
    elif args.verbose:
        logger.setLevel(logging.INFO)

    if not args.force and os.path.exists(args.output):
        logger.warning("The output image {} already exists. Exiting.".format(args.output))
        return -1

    fn, what = BOUNDARY_TERMS[args.boundary]
    boundary_term = getattr(graphcut.energy_voxel, fn)
    logger.info("Selected boundary term: " + what)

    badditional_image_data, reference_header = load(args.badditional)
    markers_image_data, _ = load(args.markers)
    fgmarkers_image_data, bgmarkers_image_data = split_marker(markers_image_data)

    if not (badditional_image_data.shape == fgmarkers_image_data.shape == bgmarkers_image_data.shape):
        logger.critical("Not all of the supplied images are of the same shape.")
        raise ValueError("Not all of the supplied images are of the same shape.")

    if args.spacing:
        spacing = get_pixel_spacing(reference_header)
        logger.info("Taking spacing of {} into account.".format(spacing))
    else:
        spacing = False

    logger.info("Building the residual lattice in HBM...")
    term_args = (badditional_image_data, spacing) if args.boundary.endswith("linear") else (badditional_image_data, args.sigma, spacing)
    gcgraph = graphcut.graph_from_voxels(fgmarkers_image_data, bgmarkers_image_data, boundary_term=boundary_term,
                                         boundary_term_args=term_args, connectivity=args.connectivity)

    logger.info("Executing min-cut...")
    maxflow = gcgraph.maxflow()
    logger.debug("Maxflow is {}".format(maxflow))

    logger.info("Applying results...")
    result_image_data = gcgraph.labels()  # == the what_segment loop of the reference, all voxels at once
    result_image_data = numpy.asarray(result_image_data).reshape(bgmarkers_image_data.shape)

    save(result_image_data.astype(numpy.bool_), args.output, reference_header, args.force)
    logger.info("Successfully terminated.")
    return 0


def getArguments(parser, argv=None):
    "Provides additional validation of the arguments collected by argparse."
    return parser.parse_args(argv)


def getParser():
    "Creates and returns the argparse parser object (reference bin/medpy_graphcut_voxel.py:197-249)."
    parser = argparse.ArgumentParser(description=__description__, formatter_class=RawTextHelpFormatter)
    parser.add_argument("sigma", type=float, help="The sigma required for the boundary terms.")
    parser.add_argument("badditional", help="The additional image required by the boundary term. See there for details.")
    parser.add_argument("markers", help="Image containing the foreground (=1) and background (=2) markers.")
    parser.add_argument("output", help="The output image containing the segmentation.")
    parser.add_argument("--boundary", default="diff_exp", choices=list(BOUNDARY_TERMS),
                        help="The boundary term to use. Note that the ones prefixed with diff_ require the original image, "
                             "while the ones prefixed with max_ require the gradient image.")
    parser.add_argument("--connectivity", type=int, default=None,
                        help="Extension: 2*ndim (default, the reference's neighbourhood) or 3**ndim-1 (8 / 26 neighbours).")
    parser.add_argument("-s", dest="spacing", action="store_true",
                        help="Set this flag to take the pixel spacing of the image into account. The spacing data will be "
                             "extracted from the baddtional image.")
    parser.add_argument("-f", dest="force", action="store_true", help="Set this flag to silently override files that exist.")
    parser.add_argument("-v", dest="verbose", action="store_true", help="Display more information.")
    parser.add_argument("-d", dest="debug", action="store_true", help="Display debug information.")
    return parser


if __name__ == "__main__":
    raise SystemExit(main())
