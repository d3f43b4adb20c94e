"""Region graph-cut command line: the reference's ``bin/medpy_graphcut_label.py`` on MI355X.

Same positional arguments, options and flow as reference bin/medpy_graphcut_label.py:77-200
(``badditional region markers output [--boundary means|stawiaski] [-f] [-v] [-d]``) plus the options of its sibling
bin/medpy_graphcut_label_w_regional.py:85-185 (``--regional none|atlas --radditional IMAGE --alpha FLOAT``: the atlas
regional term on top of the boundary term); the differences are the I/O layer
(``medpy_amd.io``: .npy / NIfTI-1 instead of SimpleITK) and the read-out (bulk ``labels()`` instead of one
``what_segment`` call per region, :150-158).
"""
import argparse
import logging
import os
from argparse import RawTextHelpFormatter

import numpy

from .. import graphcut
from ..graphcut.wrapper import ArgumentError, relabel, split_marker
from ..io import load, save

__description__ = """
Perform a binary graph cut over the REGIONS of an image (a label / watershed map) on an AMD MI355X.
Drop-in for medpy_graphcut_label.py.  With the stawiaski boundary term `badditional` is the gradient image, with the
difference of means it is the original image.  The markers image holds 1 for foreground and 2 for background seeds.
"""


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

    if args.boundary == "stawiaski":
        boundary_term = graphcut.energy_label.boundary_stawiaski
        logger.info("Selected boundary term: stawiaski")
    else:
        boundary_term = graphcut.energy_label.boundary_difference_of_means
        logger.info("Selected boundary term: difference of means")

    regional_term = graphcut.energy_label.regional_atlas if args.regional == "atlas" else False
    if regional_term and (args.radditional is None or args.alpha is None):
        raise ArgumentError("The atlas regional term needs --radditional (the probability image) and --alpha.")

    region_image_data, reference_header = load(args.region)
    badditional_image_data, _ = load(args.badditional)
    markers_image_data, _ = load(args.markers)
    radditional_image_data = load(args.radditional)[0] if regional_term else False
    fgmarkers_image_data, bgmarkers_image_data = split_marker(markers_image_data)

    if not (badditional_image_data.shape == region_image_data.shape == fgmarkers_image_data.shape == bgmarkers_image_data.shape):
        logger.critical("Not all of the supplied images are of the same shape.")
        raise ArgumentError("Not all of the supplied images are of the same shape.")

    if regional_term and not (radditional_image_data.shape == badditional_image_data.shape):
        logger.critical("Not all of the supplied images are of the same shape.")
        raise ArgumentError("Not all of the supplied images are of the same shape.")

    logger.info("Relabel input image...")
    region_image_data = relabel(region_image_data)

    logger.info("Preparing graph...")
    gcgraph = graphcut.graph_from_labels(region_image_data, fgmarkers_image_data, bgmarkers_image_data,
                                         regional_term=regional_term, boundary_term=boundary_term,
                                         regional_term_args=(radditional_image_data, args.alpha) if regional_term else False,
                                         boundary_term_args=(badditional_image_data))
    del fgmarkers_image_data, bgmarkers_image_data, badditional_image_data, radditional_image_data

    logger.info("Executing min-cut...")
    maxflow = gcgraph.maxflow()
    logger.debug("Maxflow is {}".format(maxflow))

    logger.info("Applying results...")
    mapping = numpy.concatenate([[False], gcgraph.labels()])  # entry 0 is padding: there is no region 0
    result = mapping[region_image_data]
    save(result.astype(numpy.bool_), args.output, reference_header, args.force)
    logger.info("Successfully terminated.")
    return 0


def getArguments(parser, argv=None):
    "Provides additional validation of the arguments collected by argparse."
    return parser.parse_args(argv)


def getParser():
    "Creates and returns the argparse parser object."
    parser = argparse.ArgumentParser(description=__description__, formatter_class=RawTextHelpFormatter)
    parser.add_argument("badditional", help="The additional image required by the boundary term. See there for details.")
    parser.add_argument("region", help="The region image of the image to segment.")
    parser.add_argument("markers", help="Binary image containing the foreground (=1) and background (=2) markers.")
    parser.add_argument("output", help="The output image containing the segmentation.")
    parser.add_argument("--boundary", default="stawiaski", choices=["means", "stawiaski"],
                        help="The boundary term to use. Note that difference of means (means) requires the original image, while "
                             "stawiaski requires the gradient image of the original image to be passed to badditional.")
    parser.add_argument("--regional", default="none", choices=["none", "atlas"],
                        help="The regional term to use. Note that the atlas requires to provide an atlas image.")
    parser.add_argument("--radditional", help="The additional image required by the regional term. See there for details.")
    parser.add_argument("--alpha", type=float, help="The weight of the regional term compared to the boundary term.")
    parser.add_argument("-f", dest="force", action="store_true", help="Set this flag to silently override files that exist.")
    parser.add_argument("-v", dest="verbose", action="store_true", help="Display more information.")
    parser.add_argument("-d", dest="debug", action="store_true", help="Display debug information.")
    return parser


if __name__ == "__main__":
    raise SystemExit(main())
