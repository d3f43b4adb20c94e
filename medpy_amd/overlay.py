"""Drop-in overlay: make ``import medpy.graphcut`` resolve to this package, so the reference's own command-line scripts
(bin/medpy_graphcut_voxel.py, bin/medpy_graphcut_label.py, ...) run UNMODIFIED on MI355X.

    python -m medpy_amd.overlay /path/to/MedPy/bin/medpy_graphcut_voxel.py 15 img.nii.gz markers.nii.gz out.nii.gz
    # or, inside Python, before anything imports medpy.graphcut:
    import medpy_amd.overlay; medpy_amd.overlay.install()

What ``install()`` does (reference bin/medpy_graphcut_voxel.py:20-37 is the import surface it has to satisfy):

* ``medpy.graphcut`` and its submodules ``energy_voxel``, ``energy_label``, ``generate``, ``graph``, ``wrapper``, ``write``
  become the modules of ``medpy_amd.graphcut`` (same names, signatures, plug-in protocol and exceptions);
  ``medpy.graphcut.maxflow`` -- the reference's compiled Boost.Python module (lib/maxflow/src/wrapper.cpp:59-89) -- is
  a small module whose ``GraphDouble`` is the sparse-graph solver of the HIP library;
* if a real MedPy is importable, everything else (``medpy.io``, ``medpy.core``, filters, metrics ...) stays MedPy's;
* if it is not, ``medpy``, ``medpy.core`` (``Logger``, ``ArgumentError`` ...) and ``medpy.io`` (``load``, ``save``, ``header``
  -- ``.npy`` and NIfTI-1 through ``medpy_amd.io`` instead of SimpleITK) are provided as well, enough for the graph-cut
  scripts.

There is no CPU fallback behind this: without the HIP library or without a GPU the first graph construction raises.
"""
import importlib
import logging
import runpy
import sys
import types

_SUBMODULES = ("energy_voxel", "energy_label", "generate", "graph", "wrapper", "write")


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


class _Logger(logging.Logger):
    """medpy.core.Logger (reference medpy/core/logger.py:35-138): a singleton logging.Logger writing to stdout"""
    _instance = None

    @classmethod
    def getInstance(cls):
        if cls._instance is None:
            cls._instance = cls("MedPyLogger", logging.WARNING)
            h = logging.StreamHandler(sys.stdout)
            h.setFormatter(logging.Formatter("%(asctime)s [%(levelname)-8s] %(message)s", "%d.%m.%Y %H:%M:%S"))
            cls._instance.addHandler(h)
        return cls._instance


def _core_module():
    names = ("ArgumentError", "FunctionError", "SubprocessError", "ImageTypeError", "DependencyError", "ImageLoadingError",
             "ImageSavingError", "MetaDataError")  # reference medpy/core/exceptions.py:31-70
    return _module("medpy.core", Logger=_Logger, **{n: type(n, (Exception,), {}) for n in names})


def _io_modules():
    from . import io
    header = _module("medpy.io.header", Header=io.Header, get_pixel_spacing=io.get_pixel_spacing, get_voxel_spacing=io.get_pixel_spacing,
                     get_offset=lambda hdr: hdr.offset)
    return _module("medpy.io", load=io.load, save=io.save, header=header, Header=io.Header,
                   get_pixel_spacing=io.get_pixel_spacing, get_voxel_spacing=io.get_pixel_spacing), header


def install(force_shim=False):
    """Route ``medpy.graphcut`` to ``medpy_amd.graphcut``.  Returns "patched" when a real MedPy provides the rest,
    "shim" when this module provides ``medpy``, ``medpy.core`` and ``medpy.io`` too.  Idempotent."""
    from . import graphcut
    real = None
    if not force_shim:
        try:
            real = sys.modules.get("medpy") or importlib.import_module("medpy")
            if getattr(real, "__medpy_amd_shim__", False):
                real = None
        except Exception:  # noqa: BLE001 -- not installed, or installed without its dependencies (SimpleITK ...)
            real = None
    if real is None:
        pkg = _module("medpy", __path__=[], __medpy_amd_shim__=True, __version__="0.5.2+medpy_amd")
        sys.modules["medpy"] = pkg
        pkg.core = sys.modules["medpy.core"] = _core_module()
        pkg.io, hdr = _io_modules()
        sys.modules["medpy.io"], sys.modules["medpy.io.header"] = pkg.io, hdr
    else:
        pkg = real
    sys.modules["medpy.graphcut"] = graphcut
    pkg.graphcut = graphcut
    for sub in _SUBMODULES:
        sys.modules["medpy.graphcut." + sub] = importlib.import_module("medpy_amd.graphcut." + sub)
    # the compiled module the reference's graph.py imports (from .maxflow import GraphDouble, GraphFloat, GraphInt)
    sys.modules["medpy.graphcut.maxflow"] = _module("medpy.graphcut.maxflow", GraphDouble=graphcut.GraphDouble,
                                                    GraphFloat=graphcut.GraphFloat, GraphInt=graphcut.GraphInt)
    return "shim" if real is None else "patched"


def run(script, argv=()):
    """run a MedPy command-line script (a path) under the overlay, as ``python script *argv`` would"""
    install()
    old = sys.argv
    sys.argv = [script] + list(argv)
    try:
        runpy.run_path(script, run_name="__main__")
    finally:
        sys.argv = old


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit("usage: python -m medpy_amd.overlay <MedPy script.py> [its arguments ...]")
    run(sys.argv[1], sys.argv[2:])
