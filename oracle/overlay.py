"""Run the reference's own Python (medpy.graphcut) in this container.  TEST INFRASTRUCTURE ONLY.

The reference's .py files are never copied: ``oracle/_ref/overlay/medpy`` is a tree of
symlinks into /root/reference/medpy, next to the pybind11 stand-in for the Boost.Python
``maxflow`` extension built by ``oracle/Makefile`` (``ref_pymodule.cpp``).  SimpleITK (needed
transitively by reference medpy/graphcut/wrapper.py:31 -> medpy.filter -> medpy.io) is absent
from the image, so an empty stub module is put on the path; the graph-cut code never calls it.

Only usable where /root/reference exists (this container).  Used by ``oracle/gen_golden.py``
and by the ``needs_reference`` tests.
"""
import importlib
import os
import subprocess
import sys

REF = os.environ.get("MEDPY_REFERENCE", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))
OVL = os.path.join(_HERE, "_ref", "overlay")


def reference_present():
    return os.path.isfile(os.path.join(REF, "medpy", "graphcut", "generate.py"))


def _link(src, dst):
    if os.path.islink(dst) or os.path.exists(dst):
        return
    os.symlink(src, dst)


def ensure_overlay():
    if not reference_present():
        raise RuntimeError("reference tree not present at %s" % REF)
    subprocess.check_call(["make", "-s", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    med = os.path.join(OVL, "medpy")
    os.makedirs(os.path.join(med, "graphcut"), exist_ok=True)
    os.makedirs(os.path.join(OVL, "stubs"), exist_ok=True)
    _link(os.path.join(REF, "medpy", "__init__.py"), os.path.join(med, "__init__.py"))
    for sub in ("core", "filter", "io", "features", "iterators", "metric", "neighbours", "utilities"):
        _link(os.path.join(REF, "medpy", sub), os.path.join(med, sub))
    gc = os.path.join(REF, "medpy", "graphcut")
    for f in os.listdir(gc):
        if f.endswith(".py"):
            _link(os.path.join(gc, f), os.path.join(med, "graphcut", f))
    stub = os.path.join(OVL, "stubs", "SimpleITK.py")
    if not os.path.exists(stub):
        with open(stub, "w") as fh:
            fh.write("# empty stand-in: the graph-cut path never touches SimpleITK\n")
    return OVL


def import_reference_graphcut():
    """Return the reference's ``medpy.graphcut`` package (its own .py files + compiled BK)."""
    ovl = ensure_overlay()
    for p in (os.path.join(ovl, "stubs"), ovl):
        if p not in sys.path:
            sys.path.insert(0, p)
    return importlib.import_module("medpy.graphcut")
