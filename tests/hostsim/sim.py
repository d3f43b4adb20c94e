"""ctypes wrapper around tests/hostsim/libhostsim.so (host execution of the solver's tile ops). TEST ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libhostsim.so")
_lib = None


def build():
    src = os.path.join(_HERE, "hostsim.cpp")
    deps = [src] + [os.path.join(_HERE, "..", "..", "medpy_amd", "csrc", f) for f in
                    ("mgc_tile_ops.inl", "mgc_driver.inl", "mgc_common.h")]
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(d) for d in deps):
        return
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-o", _SO, src])


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        pf = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        _lib.hostsim_solve.restype = C.c_int
        _lib.hostsim_solve.argtypes = [np.ctypeslib.ndpointer(np.int64), pf, pf, pf, pf, C.c_int, C.c_int, C.c_int, C.c_int,
                                       np.ctypeslib.ndpointer(np.uint8), np.ctypeslib.ndpointer(np.int64)]
    return _lib


STAT_NAMES = ("outer", "relabel_passes", "relabel_tiles", "phases", "discharge_tiles", "converged", "last_active", "reserved")


def solve(shape, weights, trcap, rounds=0, cycles=0, sweeps=0, max_outer=0):
    """weights: per-axis arrays for a 3-D shape (oracle layout); returns (labels[bool array], stats dict)."""
    shape = np.asarray(shape, dtype=np.int64)
    assert shape.size == 3
    ws = [np.ascontiguousarray(w, dtype=np.float64).ravel() for w in weights]
    ws = [w if w.size else np.zeros(1) for w in ws]
    tr = np.ascontiguousarray(trcap, dtype=np.float64).ravel()
    labels = np.empty(int(np.prod(shape)), np.uint8)
    stats = np.zeros(8, np.int64)
    rc = lib().hostsim_solve(shape, ws[0], ws[1], ws[2], tr, rounds, cycles, sweeps, max_outer, labels, stats)
    st = dict(zip(STAT_NAMES, stats.tolist()))
    st["rc"] = rc
    return labels.reshape(tuple(shape)), st
