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
                    ("mgc_tile_ops.inl", "mgc_tile_ops26.inl", "mgc_wave_ops.inl", "mgc_wave_ops26.inl", "mgc_dt_ops.inl", "mgc_brick_ops.inl", "mgc_driver.inl", "mgc_common.h", "mgc_terms.h")
                    if os.path.exists(os.path.join(_HERE, "..", "..", "medpy_amd", "csrc", f))]
    if os.path.exists(_SO) and all(os.path.getmtime(_SO) >= os.path.getmtime(d) for d in deps):
        return
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", "-fPIC", "-shared", "-o", _SO, src])


def build_mock_rccl():
    """tests/hostsim/libmockrccl.so: in-process stand-in for librccl (mock_rccl.cpp); needs hipcc"""
    so, src = os.path.join(_HERE, "libmockrccl.so"), os.path.join(_HERE, "mock_rccl.cpp")
    if not (os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", so, src])
    return so


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


STAT_NAMES = ("outer", "relabel_passes", "relabel_tiles", "phases", "discharge_tiles", "converged", "last_active", "reserved", "radial_cycles")


def set_wave_mode(mode):
    """which form of the hot tile operations the simulator runs: bit0 wave discharge, bit1 wave relabel (mgc_wave_ops.inl,
    one wave per tile), bit2 the wave discharge starts from exact in-tile labels, bit4 (16) the 26-neighbourhood discharge runs one wave
    per tile (mgc_wave_ops26.inl); 0 = the 512-lane workgroup forms"""
    lib().hostsim_set_wave_mode(int(mode))


def set_repeat(bits):
    """which launches of the wave discharge repeat their in-plane push steps (mgcw_discharge_impl<.., MGCW_REPEAT_MAX>): bit 0 on exact
    labels (the library's default), bit 1 during the flood on radial labels; 0 = never"""
    lib().hostsim_set_repeat(int(bits))


def set_check_exact(on):
    """after every global relabel, compare the labels with the exact distances to the sink in the residual graph (a plain
    relaxation over the volume); read the result with prof(): [40] relabels checked, [41] voxels whose label differs"""
    lib().hostsim_set_check_exact(int(bool(on)))


def prof():
    """the simulator's work counters (copied and cleared)"""
    out = np.zeros(64, np.int64)
    L = lib()
    L.hostsim_prof.argtypes = [np.ctypeslib.ndpointer(np.int64), C.c_void_p, C.c_int]
    L.hostsim_prof(out, None, 0)
    return out


def solve(shape, weights, trcap, rounds=0, cycles=0, sweeps=0, max_outer=0, wave_mode=None):
    """weights: per-axis arrays for a 3-D shape (oracle layout); returns (labels[bool array], stats dict)."""
    if wave_mode is not None:
        set_wave_mode(wave_mode)
        try:
            return solve(shape, weights, trcap, rounds, cycles, sweeps, max_outer)
        finally:
            set_wave_mode(0)
    shape = np.asarray(shape, dtype=np.int64)
    assert shape.size == 3
    ws = [np.ascontiguousarray(w, dtype=np.float64).ravel() for w in weights]
    ws = [w if w.size else np.zeros(1) for w in ws]
    tr = np.ascontiguousarray(trcap, dtype=np.float64).ravel()
    labels = np.empty(int(np.prod(shape)), np.uint8)
    stats = np.zeros(16, np.int64)
    rc = lib().hostsim_solve(shape, ws[0], ws[1], ws[2], tr, rounds, cycles, sweeps, max_outer, labels, stats)
    st = dict(zip(STAT_NAMES, stats.tolist()))
    st["rc"] = rc
    return labels.reshape(tuple(shape)), st


def first_relabel(shape, weights, trcap, use_dt):
    """the first global relabel alone: (ran_as_distance_transform, labels[tiles, 512] int32, status[tiles] uint32);
    use_dt: 0 relaxation passes, 1 distance transform, 3 transform + the radial labels of the flood phase (7: whatever the
    length of the shortest source -> sink path)"""
    shape = np.asarray(shape, dtype=np.int64)
    ws = [np.ascontiguousarray(w, dtype=np.float64).ravel() for w in weights]
    ws = [w if w.size else np.zeros(1) for w in ws]
    tr = np.ascontiguousarray(trcap, dtype=np.float64).ravel()
    nt = int(np.prod((shape + 7) // 8))
    h, st = np.zeros((nt, 512), np.int32), np.zeros(nt, np.uint32)
    L = lib()
    pf = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
    L.hostsim_first_relabel.restype = C.c_int
    L.hostsim_first_relabel.argtypes = [np.ctypeslib.ndpointer(np.int64), pf, pf, pf, pf, C.c_int,
                                        np.ctypeslib.ndpointer(np.int32), np.ctypeslib.ndpointer(np.uint32)]
    ran = L.hostsim_first_relabel(shape, ws[0], ws[1], ws[2], tr, int(use_dt), h, st)
    return bool(ran), h, st


# ------------------------------------------------------------------------------------------
# Slab handle over the host simulator: same surface as medpy_amd.slab.HipSlab, so the
# distributed schedule (medpy_amd/slab.py:solve_slabs) and the transports are tested on CPU.
# ------------------------------------------------------------------------------------------
def _solve_group(slabs, transport, params, ndir):
    """hostsim_solve_slabs: mgc_solve (mgc_driver.inl) over simulator slabs -- all slabs of the volume, or this process' one with a Python
    host transport handed over as the callbacks of an mgc_transport (medpy_amd.slab.HostTransport)"""
    from medpy_amd.slab import HostTransport
    from medpy_amd import _lib as hip
    L = lib()
    L.hostsim_solve_slabs.restype = C.c_int
    L.hostsim_solve_slabs.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.POINTER(hip.Transport), np.ctypeslib.ndpointer(np.int64),
                                      np.ctypeslib.ndpointer(np.int64)]
    p = np.zeros(8, np.int64)
    p[0] = params.get("rounds_per_relabel", 0)
    p[1] = params.get("max_cycles", 0)
    p[2] = params.get("max_sweeps", 0)
    p[3] = params.get("max_outer", 0)
    p[4] = params.get("incremental_relabel", 1)
    p[5] = params.get("exchange_passes", 0)
    p[6] = params.get("radial", -1)
    p[7] = params.get("exchange_rounds", 0)
    hs = (C.c_void_p * len(slabs))(*[s._h for s in slabs])
    st = np.zeros(16, np.int64)
    cb = HostTransport(transport) if transport is not None else None
    rc = L.hostsim_solve_slabs(hs, len(slabs), ndir, C.byref(cb.struct) if cb else None, p, st)
    if cb is not None and cb.error is not None:
        raise cb.error
    assert rc in (0, 1), "hostsim_solve_slabs failed (%d)" % rc
    out = dict(zip(STAT_NAMES, st.tolist()))
    out.update(exchanges=int(st[12]), reductions=int(st[13]), deferred_drains=int(st[9]))
    return out


class SimSlab(object):
    ndir = 6

    @staticmethod
    def solve_group(slabs, transport, params):
        return _solve_group(slabs, transport, params, 6)

    def __init__(self, global_shape, rank, nranks):
        L = lib()
        vp, i64 = C.c_void_p, C.c_int64
        L.hostsim_create.restype = vp
        L.hostsim_create.argtypes = [np.ctypeslib.ndpointer(np.int64), C.c_int, C.c_int]
        L.hostsim_destroy.argtypes = [vp]
        L.hostsim_slab_info.argtypes = [vp, np.ctypeslib.ndpointer(np.int64)]
        pf = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        L.hostsim_load.argtypes = [vp, pf, pf, pf, pf]
        L.hostsim_solver_op.argtypes = [vp, C.c_int, i64, i64, i64, i64]
        L.hostsim_read_counts.argtypes = [vp, np.ctypeslib.ndpointer(np.int32)]
        L.hostsim_halo_bytes.argtypes = [vp, C.c_int, C.POINTER(i64)]
        L.hostsim_halo_pack.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
        L.hostsim_halo_unpack.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_uint32, C.c_int]
        L.hostsim_labels.argtypes = [vp, np.ctypeslib.ndpointer(np.uint8)]
        self._L = L
        self._h = L.hostsim_create(np.asarray(global_shape, dtype=np.int64), rank, nranks)
        assert self._h, "cannot cut the volume into that many slabs"
        info = np.zeros(8, np.int64)
        L.hostsim_slab_info(self._h, info)
        self.plane0, self.plane1, self.own0, self.own1 = (int(v) for v in info[:4])
        self.has_lo, self.has_hi = bool(info[4]), bool(info[5])
        self.local_shape = (self.plane1 - self.plane0, int(global_shape[1]), int(global_shape[2]))
        self.rank, self.nranks = rank, nranks

    def close(self):
        if self._h:
            self._L.hostsim_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def load(self, weights_global, trcap_global):
        """slice the LOCAL planes out of the global per-axis weights / t-links and load them"""
        a, b = self.plane0, self.plane1
        w0 = np.ascontiguousarray(weights_global[0][a:b - 1], dtype=np.float64).ravel()
        w1 = np.ascontiguousarray(weights_global[1][a:b], dtype=np.float64).ravel()
        w2 = np.ascontiguousarray(weights_global[2][a:b], dtype=np.float64).ravel()
        tr = np.ascontiguousarray(np.asarray(trcap_global).reshape((-1,) + self.local_shape[1:])[a:b], dtype=np.float64).ravel()
        pad = lambda w: w if w.size else np.zeros(1)
        self._L.hostsim_load(self._h, pad(w0), pad(w1), pad(w2), tr)

    def op(self, op, a0=0, a1=0, a2=0, a3=0):
        assert self._L.hostsim_solver_op(self._h, int(op), int(a0), int(a1), int(a2), int(a3)) == 0

    def read_counts(self):
        out = np.zeros(32, np.int32)
        self._L.hostsim_read_counts(self._h, out)
        return out

    def set_halo_max(self, n):
        """record slots of a compacted border message (MgcLattice::halo_max_rec): small values force the deferral path"""
        self._L.hostsim_set_halo_max.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self._L.hostsim_set_halo_max(self._h, 6, int(n))

    def halo_bytes(self, kind):
        n = C.c_int64(0)
        self._L.hostsim_halo_bytes(self._h, int(kind), C.byref(n))
        return n.value

    def halo_pack(self, side, kind, buf, on_device=False):
        self._L.hostsim_halo_pack(self._h, int(side), int(kind), C.c_void_p(buf.ctypes.data), 0)

    def halo_unpack(self, side, kind, buf, epoch, lst, on_device=False):
        self._L.hostsim_halo_unpack(self._h, int(side), int(kind), C.c_void_p(buf.ctypes.data), 0, int(epoch), int(lst))

    def finish(self):
        out = np.empty(int(np.prod(self.local_shape)), np.uint8)
        self._L.hostsim_labels(self._h, out)
        return out.reshape(self.local_shape)[self.own0 - self.plane0:self.own1 - self.plane0].astype(np.bool_), 0.0


def weights26(shape, weights_by_offset):
    """{offset: array (NaN where no neighbour)} for the 13 forward offsets -> dense (26,) + shape array of arc
    capacities in mgc26_offset order (reverse arcs mirrored: the capacities are symmetric)."""
    shape = tuple(int(v) for v in shape)
    w = np.zeros((26,) + shape)
    for o, arr in weights_by_offset.items():
        code = (o[0] + 1) * 9 + (o[1] + 1) * 3 + (o[2] + 1)
        d = code if code < 13 else code - 1
        fwd = np.nan_to_num(np.asarray(arr, dtype=np.float64), nan=0.0)
        w[d] = fwd
        # the reverse arc of (p, p+o) leaves p+o in direction -o with the same (symmetric) capacity
        src = tuple(slice(max(0, -k), shape[a] - max(0, k)) for a, k in enumerate(o))
        dst = tuple(slice(max(0, k), shape[a] - max(0, -k)) for a, k in enumerate(o))
        w[25 - d][dst] = fwd[src]
    return w


def solve26(shape, weights_by_offset, trcap, rounds=0, cycles=0, sweeps=0, max_outer=0, wave_mode=None):
    if wave_mode is not None:  # bit 4 (16): the discharge runs one wave per tile (mgc_wave_ops26.inl)
        set_wave_mode(wave_mode)
        try:
            return solve26(shape, weights_by_offset, trcap, rounds, cycles, sweeps, max_outer)
        finally:
            set_wave_mode(0)
    """26-neighbourhood: weights_by_offset = {offset: array (NaN where no neighbour)} for the 13 forward offsets
    (oracle/energy_numpy.py:boundary_weights_offsets).  Returns (labels, stats)."""
    L = lib()
    pf = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
    L.hostsim_solve26.restype = C.c_int
    L.hostsim_solve26.argtypes = [np.ctypeslib.ndpointer(np.int64), pf, pf, C.c_int, C.c_int, C.c_int, C.c_int,
                                  np.ctypeslib.ndpointer(np.uint8), np.ctypeslib.ndpointer(np.int64)]
    shape = tuple(int(v) for v in shape)
    n = int(np.prod(shape))
    w = weights26(shape, weights_by_offset)
    labels = np.empty(n, np.uint8)
    stats = np.zeros(16, np.int64)
    rc = L.hostsim_solve26(np.asarray(shape, np.int64), np.ascontiguousarray(w).ravel(), np.ascontiguousarray(trcap, dtype=np.float64).ravel(),
                           rounds, cycles, sweeps, max_outer, labels, stats)
    st = dict(zip(STAT_NAMES, stats.tolist()))
    st["rc"] = rc
    return labels.reshape(shape), st


class SimSlab26(object):
    """26-neighbourhood slab over the host simulator (same surface as SimSlab / medpy_amd.slab.HipSlab)."""
    ndir = 26

    @staticmethod
    def solve_group(slabs, transport, params):
        return _solve_group(slabs, transport, params, 26)

    def __init__(self, global_shape, rank, nranks):
        L = lib()
        vp, i64 = C.c_void_p, C.c_int64
        L.hostsim26_create.restype = vp
        L.hostsim26_create.argtypes = [np.ctypeslib.ndpointer(np.int64), C.c_int, C.c_int]
        L.hostsim26_destroy.argtypes = [vp]
        L.hostsim26_slab_info.argtypes = [vp, np.ctypeslib.ndpointer(np.int64)]
        pf = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
        L.hostsim26_load.argtypes = [vp, pf, pf]
        L.hostsim26_solver_op.argtypes = [vp, C.c_int, i64, i64, i64, i64]
        L.hostsim26_read_counts.argtypes = [vp, np.ctypeslib.ndpointer(np.int32)]
        L.hostsim26_halo_bytes.argtypes = [vp, C.c_int, C.POINTER(i64)]
        L.hostsim26_halo_pack.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int]
        L.hostsim26_halo_unpack.argtypes = [vp, C.c_int, C.c_int, vp, C.c_int, C.c_uint32, C.c_int]
        L.hostsim26_labels.argtypes = [vp, np.ctypeslib.ndpointer(np.uint8)]
        self._L = L
        self._h = L.hostsim26_create(np.asarray(global_shape, dtype=np.int64), rank, nranks)
        assert self._h, "cannot cut the volume into that many slabs"
        info = np.zeros(8, np.int64)
        L.hostsim26_slab_info(self._h, info)
        self.plane0, self.plane1, self.own0, self.own1 = (int(v) for v in info[:4])
        self.has_lo, self.has_hi = bool(info[4]), bool(info[5])
        self.local_shape = (self.plane1 - self.plane0, int(global_shape[1]), int(global_shape[2]))
        self.rank, self.nranks = rank, nranks

    def close(self):
        if self._h:
            self._L.hostsim26_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def load(self, w26_global, trcap_global):
        """w26_global: (26,) + global shape (weights26); the LOCAL planes are sliced out"""
        a, b = self.plane0, self.plane1
        w = np.ascontiguousarray(w26_global[:, a:b], dtype=np.float64).ravel()
        tr = np.ascontiguousarray(np.asarray(trcap_global).reshape((-1,) + self.local_shape[1:])[a:b], dtype=np.float64).ravel()
        self._L.hostsim26_load(self._h, w, tr)

    def op(self, op, a0=0, a1=0, a2=0, a3=0):
        assert self._L.hostsim26_solver_op(self._h, int(op), int(a0), int(a1), int(a2), int(a3)) == 0

    def read_counts(self):
        out = np.zeros(32, np.int32)
        self._L.hostsim26_read_counts(self._h, out)
        return out

    def set_halo_max(self, n):
        self._L.hostsim_set_halo_max.argtypes = [C.c_void_p, C.c_int, C.c_int]
        self._L.hostsim_set_halo_max(self._h, 26, int(n))

    def halo_bytes(self, kind):
        n = C.c_int64(0)
        self._L.hostsim26_halo_bytes(self._h, int(kind), C.byref(n))
        return n.value

    def halo_pack(self, side, kind, buf, on_device=False):
        self._L.hostsim26_halo_pack(self._h, int(side), int(kind), C.c_void_p(buf.ctypes.data), 0)

    def halo_unpack(self, side, kind, buf, epoch, lst, on_device=False):
        self._L.hostsim26_halo_unpack(self._h, int(side), int(kind), C.c_void_p(buf.ctypes.data), 0, int(epoch), int(lst))

    def finish(self):
        out = np.empty(int(np.prod(self.local_shape)), np.uint8)
        self._L.hostsim26_labels(self._h, out)
        return out.reshape(self.local_shape)[self.own0 - self.plane0:self.own1 - self.plane0].astype(np.bool_), 0.0
