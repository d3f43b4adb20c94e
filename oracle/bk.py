"""ctypes front-end for the two CPU max-flow checkers.  TEST INFRASTRUCTURE ONLY.

* ``kind="ref"``  -> ``oracle/_ref/libbkref.so``: the UNMODIFIED reference BK solver
  (reference lib/maxflow/src/{graph.h,graph.cpp,maxflow.cpp}) compiled in place by
  ``oracle/Makefile`` behind bulk C entry points (``oracle/ref_bulk.cpp``).
* ``kind="port"`` -> ``oracle/libbkport.so``: this repository's C restatement
  (``oracle/bk_maxflow.c``), pinned bit-for-bit against ``ref`` by
  ``tests/test_oracle_port_vs_ref.py``.

Both expose the same calls, mirroring what the reference drives per edge / node from
Python (graph.py:438-440, 496-498; bin/medpy_graphcut_voxel.py:172-181).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATHS = {
    "ref": os.path.join(_HERE, "_ref", "libbkref.so"),
    "port": os.path.join(_HERE, "libbkport.so"),
}
_LIBS = {}

SOURCE, SINK = 0, 1  # termtype, reference graph.h:57-61


def build(verbose=False):
    """Compile the checkers (gcc/g++ only).  ``ref`` is built only where /root/reference exists."""
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(["make", "-s", "-C", _HERE, "port"], stdout=out)
    subprocess.check_call(["make", "-s", "-C", _HERE, "ref"], stdout=out)


def available(kind):
    return os.path.exists(_PATHS[kind])


def best_kind():
    """``ref`` when the compiled reference travelled with the snapshot, else ``port``."""
    return "ref" if available("ref") else "port"


def _lib(kind):
    if kind in _LIBS:
        return _LIBS[kind]
    if not available(kind):
        build()
    lib = C.CDLL(_PATHS[kind])
    p = "bk%s_" % kind
    i64, dbl, vp = C.c_int64, C.c_double, C.c_void_p
    pi64 = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
    pf64 = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
    pu8 = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
    sig = {
        "create": (vp, [i64, i64]),
        "destroy": (None, [vp]),
        "sum_edges": (None, [vp, i64, pi64, pi64, pf64, pf64]),
        "add_edges": (None, [vp, i64, pi64, pi64, pf64, pf64]),
        "add_tweights": (None, [vp, i64, C.c_void_p, pf64, pf64]),
        "sum_lattice": (None, [vp, C.c_int, pi64, C.POINTER(C.c_void_p)]),
        "maxflow": (dbl, [vp]),
        "labels": (None, [vp, i64, pu8]),
        "what_segment": (C.c_int, [vp, i64]),
        "get_trcap": (dbl, [vp, i64]),
        "get_edge": (dbl, [vp, i64, i64]),
        "get_node_num": (i64, [vp]),
        "get_arc_num": (i64, [vp]),
        "export": (None, [vp, np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS"), np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS"),
                          pf64, pf64]),
    }
    ns = {}
    for name, (res, args) in sig.items():
        f = getattr(lib, p + name)
        f.restype, f.argtypes = res, args
        ns[name] = f
    _LIBS[kind] = ns
    return ns


class BKGraph:
    """GraphDouble-shaped handle (reference wrapper.cpp:59-89) with bulk setters."""

    def __init__(self, nodes, edges, kind=None):
        self.kind = kind or best_kind()
        self._f = _lib(self.kind)
        self._h = self._f["create"](int(nodes), int(edges))
        if not self._h:
            raise MemoryError("oracle: graph too large for 32-bit ids (reference graph.h:62,82)")
        self.nodes = int(nodes)

    def __del__(self):
        if getattr(self, "_h", None):
            self._f["destroy"](self._h)
            self._h = None

    @staticmethod
    def _arr(a, dt):
        return np.ascontiguousarray(a, dtype=dt)

    def sum_edges(self, i, j, cap, rev=None):
        i, j = self._arr(i, np.int64), self._arr(j, np.int64)
        cap = self._arr(cap, np.float64)
        rev = cap if rev is None else self._arr(rev, np.float64)
        self._f["sum_edges"](self._h, i.size, i, j, cap, rev)

    def add_edges(self, i, j, cap, rev=None):
        i, j = self._arr(i, np.int64), self._arr(j, np.int64)
        cap = self._arr(cap, np.float64)
        rev = cap if rev is None else self._arr(rev, np.float64)
        self._f["add_edges"](self._h, i.size, i, j, cap, rev)

    def add_tweights(self, idx, src, snk):
        src, snk = self._arr(src, np.float64), self._arr(snk, np.float64)
        if idx is None:
            self._f["add_tweights"](self._h, src.size, None, src, snk)
        else:
            idx = self._arr(idx, np.int64)
            self._f["add_tweights"](self._h, idx.size, idx.ctypes.data, src, snk)

    def sum_lattice(self, shape, weights):
        """weights[d]: the per-axis array of __skeleton_base (energy_voxel.py:644-658), any shape, C order."""
        shape = self._arr(shape, np.int64)
        ws = [self._arr(w, np.float64).ravel() for w in weights]
        ptrs = (C.c_void_p * len(ws))(*[w.ctypes.data for w in ws])
        self._f["sum_lattice"](self._h, len(ws), shape, ptrs)

    def maxflow(self):
        return self._f["maxflow"](self._h)

    def labels(self):
        out = np.empty(self.nodes, np.uint8)
        self._f["labels"](self._h, self.nodes, out)
        return out

    def what_segment(self, i):
        return self._f["what_segment"](self._h, int(i))

    def get_trcap(self, i):
        return self._f["get_trcap"](self._h, int(i))

    def get_edge(self, i, j):
        return self._f["get_edge"](self._h, int(i), int(j))

    def get_node_num(self):
        return self._f["get_node_num"](self._h)

    def get_arc_num(self):
        return self._f["get_arc_num"](self._h)

    def export(self):
        """(tail, head, rcap, trcap): every arc in allocation order (sisters adjacent) with its residual capacity, and the
        residual t-links (> 0: source -> node, < 0: node -> sink) -- the residual graph after maxflow()"""
        na = int(self.get_arc_num())
        tail, head = np.empty(na, np.int32), np.empty(na, np.int32)
        rcap, trcap = np.empty(na, np.float64), np.empty(self.nodes, np.float64)
        self._f["export"](self._h, tail, head, rcap, trcap)
        return tail, head, rcap, trcap
