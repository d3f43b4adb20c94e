"""Exact Z-slab decomposition of one voxel graph cut over several GPUs.

No reference counterpart: the reference is single-process, and its only splitter
(``graphcut_split``, reference medpy/graphcut/wrapper.py:72-204) is approximate and
label-based.  Here the split is exact: the volume is cut along axis 0 into slabs of whole
8-plane tile layers, one slab per GPU; a slab boundary is an ordinary tile face whose neighbour
lives on another GPU.  After every relabel pass / colour phase each slab packs the labels of
its border voxels and the flow it pushed across the boundary (``mgc_halo_pack``), the packed
border travels to the neighbour rank (RCCL send/recv over xGMI, driven by the library itself,
or an in-process loopback), and is unpacked into the ghost tiles there (``mgc_halo_unpack``).
Because region discharge only ever reads a neighbour tile's labels and outbox as of that
tile's last discharge, the distributed run performs exactly the single-GPU computation:
labels are bit-identical to the single-GPU (and hence the reference's) labels.

The schedule itself is the single handle's (``mgc_solve``, ``medpy_amd/csrc/mgc_driver.inl``) with the borders exchanged at its
hook points; it lives in the library (``mgc_solve_slabs``) and, as the same source, in the host simulator of the CPU test tier.
This module holds the slab handle, the layouts (all slabs in one process / one per process over RCCL / one per process over a host
transport: two processes over gloo or over a directory of files in tests/test_slab_*.py) and the helpers around a solve.
"""
import numpy as np

(OP_ABSORB_ALL, OP_FILL_INF, OP_ZERO_COUNT, OP_RELABEL_ALL, OP_RELABEL_LIST, OP_ACTIVATE, OP_DISCHARGE, OP_SUSPECT_PASS,
 OP_RESET_SUSPECT) = range(9)  # mgc_solver_op: single launches for profiling tools (tools/gpu_slab_ops_profile.py); no schedule uses them


class LoopbackExchange(object):
    """All slabs of the volume live in this process (one GPU time-multiplexed, or the host simulator): their borders move inside the
    library / the simulator (``solve_slabs``); what is left here are the sums over the local slabs that the helpers below ask for."""

    def __init__(self, slabs):
        self.slabs = list(slabs)

    def allreduce_sum(self, values):
        """values: one number (or vector) per local slab -> global sum"""
        return np.sum(np.asarray(values, dtype=np.float64), axis=0)

    def allreduce_max(self, values):
        """values: one vector per local slab -> element-wise global maximum"""
        return np.max(np.asarray(values, dtype=np.float64), axis=0)


class StoreExchange(object):
    """One slab per process; the packed borders travel through host buffers and the out-of-band store
    (medpy_amd.rendezvous.FileStore: a file per message).  A DEVELOPMENT transport -- ranks may share one GPU, nothing here is
    fast -- that exercises the multi-process schedule where no second GPU (or no RCCL) is to be had: the CPU test tier over the
    host simulator and the reduced-size multi-rank run of bench.py.  The production transport is RcclExchange below."""

    def __init__(self, slab, store):
        self.store, self.slabs = store, [slab]
        self.rank, self.world = store.rank, store.world

    # -- what HostTransport hands to the library as the callbacks of an mgc_transport
    def xchg(self, lo, hi):
        for peer, data in ((self.rank - 1, lo), (self.rank + 1, hi)):
            if data is not None:
                self.store.send(peer, data)
        return (self.store.recv(self.rank - 1, len(lo)) if lo is not None else None,
                self.store.recv(self.rank + 1, len(hi)) if hi is not None else None)

    def allreduce_i64(self, a, op):
        a = np.asarray(a, dtype=np.float64)  # (counters and distances: far below 2^53)
        out = self.store.allreduce(a, "sum") if op == 0 else -np.asarray(self.store.allreduce(-a, "max"))
        return np.asarray(out).reshape(-1).astype(np.int64)

    def send(self, side, data):
        self.store.send(self.rank + (1 if side else -1), data)

    def recv(self, side, nbytes):
        return self.store.recv(self.rank + (1 if side else -1), nbytes)

    def allreduce_sum(self, values):
        out = self.store.allreduce(np.sum(np.asarray(values, dtype=np.float64), axis=0), "sum")
        return out if out.size > 1 else float(out[0])

    def allreduce_max(self, values):
        return self.store.allreduce(np.max(np.asarray(values, dtype=np.float64), axis=0), "max")


class RcclExchange(object):
    """One slab per process, borders moved by the library itself: pack -> grouped ncclSend/ncclRecv with
    rank-1 / rank+1 -> unpack, stream-ordered in HBM (``mgc_halo_exchange``); counters reduced with ncclAllReduce, the carry planes
    of the distance transforms by ncclSend / ncclRecv.  ``store`` (medpy_amd.rendezvous.FileStore, or anything with ``rank``, ``world``,
    ``broadcast(bytes, src, nbytes)`` and ``allreduce(array, op)``) is only the out-of-band channel that hands rank 0's
    128-byte RCCL id to the other ranks and sums a few host scalars at the end -- no PyTorch anywhere."""

    def __init__(self, slab, store):
        self.store = store
        self.slabs = [slab]
        self.rank, self.world = store.rank, store.world
        uid = slab.comm_unique_id() if self.rank == 0 else b""
        slab.comm_init(store.broadcast(uid, src=0, nbytes=128))

    native = True  # the slab has its own channel (mgc_comm_init): no callbacks

    def allreduce_sum(self, values):
        out = self.store.allreduce(np.sum(np.asarray(values, dtype=np.float64), axis=0), "sum")
        return out if out.size > 1 else float(out[0])

    def allreduce_max(self, values):
        return self.store.allreduce(np.max(np.asarray(values, dtype=np.float64), axis=0), "max")


def sync_image_range(slabs, ex):
    """The *_linear boundary terms normalise by the intensity range of the WHOLE volume (energy_voxel.py:101, 174-176):
    reduce the slabs' local {min, max, max|.|} over all ranks and hand every slab the global triple.  Call between
    set_boundary() and build(); a no-op for slabs without this notion (the host simulator is handed finished weights)."""
    local = []
    for s in slabs:
        if not hasattr(s, "image_range"):
            return
        mn, mx, ma = s.image_range()
        local.append([-mn, mx, ma])
    g = np.asarray(ex.allreduce_max(local), dtype=np.float64).reshape(-1)
    for s in slabs:
        s.set_image_range(-g[0], g[1], g[2])


def sync_boundary_table(slabs, ex):
    """One decision for the whole volume: the exponential / power term goes by table iff EVERY slab holds whole numbers only and the
    GLOBAL value range is small enough (graph.py:boundary_table, reference energy_voxel.py:226-236, 290-300, 444-452, 506-513).
    Call between set_boundary() and build(), like sync_image_range(); a no-op for slabs without images (the host simulator)."""
    facts = []
    for s in slabs:
        if not hasattr(s, "table_facts"):
            return
        f = s.table_facts()
        facts.append([0.0, -np.inf, -np.inf] if f is None else [1.0 if f[0] else 0.0, -f[1], f[2]])
    # min over "whole numbers only" = -max(-x); max over (-min) and max
    g = np.asarray(ex.allreduce_max([[-f[0], f[1], f[2]] for f in facts]), dtype=np.float64).reshape(-1)
    if -g[0] < 1.0 or not np.isfinite(g[1]) or not np.isfinite(g[2]):
        return
    for s in slabs:
        s.set_boundary_table(-g[1], g[2])


def solve_slabs(slabs, ex=None, rounds_per_relabel=None, max_cycles=None, max_sweeps=None, max_outer=None, check_rounds=None, relabel_batch=None,
                incremental_relabel=True, exchange_every=None, exchange_rounds=None, radial=None):
    """Drives the local slabs to a maximum preflow.  Returns a stats dict (global numbers).

    There is ONE schedule, and it is not here: ``mgc_solve`` (medpy_amd/csrc/mgc_driver.inl) -- the single handle's own, with the
    borders exchanged at its hook points: first relabel by distance transform carried across the slab borders, flood phase on radial
    labels, incremental relabels whose suspect closure crosses the borders.  The library runs it for HipSlab handles
    (``mgc_solve_slabs``), the host simulator of the CPU test tier for its slabs (the same source, ``hostsim_solve_slabs``).  What
    this function chooses is the layout:

    * ``slabs`` = every slab of the volume (time-multiplexed on one GPU, or simulator slabs): borders move between the slabs' own
      buffers; ``ex`` is ignored (a ``LoopbackExchange`` by tradition);
    * one slab and ``ex`` = ``RcclExchange``: borders and reductions over RCCL / xGMI inside the library;
    * one slab and a host transport (``StoreExchange``; the gloo transport of the tests): borders through the transport's
      ``xchg`` / ``allreduce_i64`` / ``send`` / ``recv``, handed to the library as the callbacks of an ``mgc_transport``.

    ``None`` leaves a parameter at the library's default for the slab's neighbourhood."""
    params = {"rounds_per_relabel": rounds_per_relabel, "max_cycles": max_cycles, "max_sweeps": max_sweeps, "max_outer": max_outer,
              "check_rounds": check_rounds, "relabel_batch": relabel_batch, "incremental_relabel": int(bool(incremental_relabel)),
              "exchange_passes": exchange_every, "exchange_rounds": exchange_rounds, "radial": radial}
    transport = None
    if len(slabs) == 1 and ex is not None and not getattr(ex, "native", False) and not isinstance(ex, LoopbackExchange):
        transport = ex
    return slabs[0].solve_group(slabs, transport, {k: v for k, v in params.items() if v is not None})


class HostTransport(object):
    """the four callbacks of an ``mgc_transport`` (include/medpy_hip.h) over a Python transport ``ex`` with
    ``xchg(lo, hi) -> (lo', hi')`` (bytes or None per side), ``allreduce_i64(array, op)`` (op 0 sum / 1 min, returns the array),
    ``send(side, bytes)`` and ``recv(side, nbytes) -> bytes``.  Keeps the ctypes thunks alive for as long as it lives."""

    def __init__(self, ex):
        import ctypes as C
        from . import _lib
        self.error = None

        def guard(fn):
            def wrapped(*a):
                try:
                    fn(*a)
                    return 0
                except Exception as e:  # noqa: BLE001  (an exception must not unwind through the C frames: report and fail the call)
                    self.error = e
                    return _lib.ERR_INVALID
            return wrapped

        def exchange(_ctx, send_lo, recv_lo, send_hi, recv_hi, n):
            lo, hi = ex.xchg(C.string_at(send_lo, n) if send_lo else None, C.string_at(send_hi, n) if send_hi else None)
            if recv_lo:
                C.memmove(recv_lo, lo, n)
            if recv_hi:
                C.memmove(recv_hi, hi, n)

        def allreduce(_ctx, v, n, op):
            a = np.ctypeslib.as_array(v, shape=(n,))
            a[:] = np.asarray(ex.allreduce_i64(a.copy(), int(op)), dtype=np.int64)

        def send(_ctx, side, buf, n):
            ex.send(int(side), C.string_at(buf, n))

        def recv(_ctx, side, buf, n):
            C.memmove(buf, ex.recv(int(side), int(n)), n)

        self._thunks = (_lib.XCHG_FN(guard(exchange)), _lib.ALLREDUCE_FN(guard(allreduce)), _lib.SEND_FN(guard(send)), _lib.RECV_FN(guard(recv)))
        self.struct = _lib.Transport(None, *self._thunks)


class HipSlab(object):
    """One Z-slab of a volume on one MI355X (C ABI: mgc_create_slab ... mgc_finish)."""

    def __init__(self, global_shape, rank, nranks, device=0, connectivity=6):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib
        lib = _lib.load()
        if _lib.device_count() < 1:
            raise _lib.MedpyHipError(_lib.ERR_NO_DEVICE, "no HIP device visible; medpy_amd has no CPU fallback")
        shp = (C.c_int64 * 3)(*[int(v) for v in global_shape])
        h = C.c_void_p()
        if connectivity not in (6, 26):
            raise ValueError("slabs are cut from 3-D volumes: connectivity is 6 or 26")
        self.ndir = int(connectivity)
        rc = lib.mgc_create_slab(3, shp, self.ndir, int(device), int(rank), int(nranks), C.byref(h))
        self._h = h if h.value else None
        if rc != _lib.OK:
            msg = (lib.mgc_last_error(self._h) or b"").decode()
            self.close()
            raise _lib.MedpyHipError(rc, msg)
        info = (C.c_int64 * 8)()
        self._call("mgc_slab_info", info)
        self.plane0, self.plane1, self.own0, self.own1 = int(info[0]), int(info[1]), int(info[2]), int(info[3])
        self.has_lo, self.has_hi = bool(info[4]), bool(info[5])
        self.local_shape = (self.plane1 - self.plane0, int(global_shape[1]), int(global_shape[2]))
        self.rank, self.nranks = rank, nranks
        _lib.apply_env_params(self._h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.load().mgc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        self._lib.check(self._h, getattr(self._lib.load(), name)(self._h, *args))

    # -- inputs: LOCAL planes [plane0, plane1) of the global arrays
    def set_boundary(self, term, image_local, sigma, spacing=False):
        C, _lib = self._C, self._lib
        image = np.ascontiguousarray(image_local)
        assert image.shape == self.local_shape, (image.shape, self.local_shape)
        if image.dtype not in _lib.DTYPE_IDS:
            image = image.astype(np.float64)
        sp = (C.c_double * 3)(*[float(v) for v in spacing]) if spacing else None
        self._call("mgc_set_boundary", _lib.TERM_IDS[term], _lib.ptr(image), _lib.DTYPE_IDS[image.dtype],
                   float(sigma) if sigma is not None else 0.0, sp)
        # The term by table (integer-valued images, graph.py:boundary_table) is decided for the WHOLE volume, not per slab: one slab
        # on the table and its neighbour on the device's own exp / pow would put weights up to 2 ulp apart on either side of the
        # border, and the slab labels would no longer be those of the single handle.  sync_boundary_table() hands out the table
        # after every rank has said what its planes hold; until then the device functions stand.
        self._term, self._sigma = term, sigma
        from .graphcut.graph import image_table_facts
        self._table_facts = image_table_facts(term, image)

    def table_facts(self):
        """(whole numbers only?, min, max) of the local planes, or None when the term has no table"""
        return getattr(self, "_table_facts", None)

    def set_boundary_table(self, lo, hi):
        from .graphcut.graph import boundary_table_for_range
        table = boundary_table_for_range(self._term, self._sigma, lo, hi)
        if table is not None:
            self._call("mgc_set_boundary_lut", self._lib.ptr(table), table.size)

    def image_range(self):
        out = np.zeros(3, dtype=np.float64)
        self._call("mgc_get_image_range", self._lib.ptr(out))
        return float(out[0]), float(out[1]), float(out[2])

    def set_image_range(self, mn, mx, maxabs):
        v = np.asarray([mn, mx, maxabs], dtype=np.float64)
        self._call("mgc_set_image_range", self._lib.ptr(v))

    def set_markers(self, fg_local, bg_local):
        def as_bytes(m):  # (a contiguous bool / byte array IS what the library reads: no copy)
            m = np.asarray(m)
            return m.view(np.uint8) if (m.dtype in (np.bool_, np.uint8) and m.flags.c_contiguous) else np.ascontiguousarray(m, dtype=np.bool_).view(np.uint8)
        fg, bg = as_bytes(fg_local), as_bytes(bg_local)
        self._call("mgc_set_markers", self._lib.ptr(fg), self._lib.ptr(bg))

    def set_regional(self, prob_local, alpha):
        prob = np.ascontiguousarray(prob_local)
        if prob.dtype not in (np.float32, np.float64):
            prob = prob.astype(np.float64)
        self._call("mgc_set_regional_probability", self._lib.ptr(prob), self._lib.DTYPE_IDS[prob.dtype], float(alpha))

    def build(self):
        self._call("mgc_build")

    # -- the stepwise solver surface used by solve_slabs()
    def op(self, op, a0=0, a1=0, a2=0, a3=0):
        self._call("mgc_solver_op", int(op), int(a0), int(a1), int(a2), int(a3))

    def read_counts(self):
        out = np.zeros(32, dtype=np.int32)
        self._call("mgc_read_counts", self._lib.ptr(out))
        return out

    def set_param(self, name, value):
        self._call("mgc_set_param", name.encode(), int(value))

    @staticmethod
    def solve_group(slabs, transport, params):
        """mgc_solve_slabs over ``slabs`` (all slabs of the volume, or this rank's one); ``transport``: None, or a Python host transport"""
        import ctypes as C
        from . import _lib
        lib = _lib.load()
        for name, value in params.items():
            slabs[0].set_param(name, value)  # (the group runs on the parameters of its first slab)
        hs = (C.c_void_p * len(slabs))(*[s._h for s in slabs])
        st = _lib.SlabStats()
        cb = HostTransport(transport) if transport is not None else None
        rc = lib.mgc_solve_slabs(hs, len(slabs), C.byref(cb.struct) if cb else None, C.byref(st))
        if cb is not None and cb.error is not None:
            raise cb.error
        if rc not in (_lib.OK, _lib.ERR_NOT_CONVERGED):  # (max_outer exhausted: the stats say converged = 0, as they always did)
            _lib.check(slabs[0]._h, rc)
        out = st.as_dict()
        out["radial_cycles"] = int(st.reserved[0])
        return out

    def halo_bytes(self, kind):
        n = self._C.c_int64(0)
        self._call("mgc_halo_bytes", int(kind), self._C.byref(n))
        return n.value

    def _ptr(self, buf, on_device):
        return self._C.c_void_p(int(buf)) if on_device else self._lib.ptr(buf)

    def halo_pack(self, side, kind, buf, on_device=False):
        self._call("mgc_halo_pack", int(side), int(kind), self._ptr(buf, on_device), int(bool(on_device)))

    def halo_unpack(self, side, kind, buf, epoch, lst, on_device=False):
        self._call("mgc_halo_unpack", int(side), int(kind), self._ptr(buf, on_device), int(bool(on_device)), int(epoch), int(lst))

    # -- native transport (RCCL): see RcclExchange
    def comm_unique_id(self):
        buf = np.zeros(128, dtype=np.uint8)
        rc = self._lib.load().mgc_comm_unique_id(self._lib.ptr(buf))
        self._lib.check(None, rc)
        return buf.tobytes()

    def comm_init(self, id_bytes):
        buf = np.frombuffer(id_bytes, dtype=np.uint8).copy()
        self._call("mgc_comm_init", self._lib.ptr(buf))

    def exchange(self, kind, epoch, lst):
        self._call("mgc_halo_exchange", int(kind), int(epoch), int(lst))

    def allreduce_counts(self):
        out = np.zeros(32, dtype=np.int64)
        self._call("mgc_allreduce_counts", self._lib.ptr(out))
        return out

    def finish_device(self):
        """labels (left in HBM) + this slab's part of the cut value"""
        f = self._C.c_double(0.0)
        self._call("mgc_finish", self._C.byref(f))
        return f.value

    def finish(self):
        """labels of the OWNED planes (bool, False where what_segment == SINK) and this slab's part of the cut value"""
        f = self._C.c_double(self.finish_device())
        out = np.empty(int(np.prod(self.local_shape)), dtype=np.uint8)
        self._call("mgc_labels", self._lib.ptr(out))
        lab = out.reshape(self.local_shape)[self.own0 - self.plane0:self.own1 - self.plane0].astype(np.bool_)
        return lab, f.value

    def stats(self):
        st = self._lib.Stats()
        self._call("mgc_get_stats", self._C.byref(st))
        return st.as_dict()

    def validate(self):
        """this slab's part of the max-flow invariants (mgc_validate); sum the dicts over the slabs (validate_slabs)"""
        v = self._lib.Validation()
        self._call("mgc_validate", self._C.byref(v))
        return v.as_dict()


def validate_slabs(slabs, ex):
    """Invariants of the maximum preflow of a volume cut into slabs: every rank counts over its own planes, the counts
    and the two flow values are summed over all ranks (the conservation errors: maximum).  Returns the global dict;
    medpy_amd._lib.assert_valid(dict) raises on a violation.  The only check there is for volumes no CPU oracle reaches."""
    from . import _lib
    parts = [s.validate() for s in slabs]
    keys = [k for k in parts[0] if k not in ("max_pair_error", "max_node_error")]
    sums = np.asarray(ex.allreduce_sum([[float(p[k]) for k in keys] for p in parts]), dtype=np.float64).reshape(-1)
    mx = np.asarray(ex.allreduce_max([[p["max_pair_error"], p["max_node_error"]] for p in parts]), dtype=np.float64).reshape(-1)
    out = {k: (float(v) if k in ("flow_into_sink", "cut_capacity", "flow_constant", "sink_capacity_used") else int(round(v))) for k, v in zip(keys, sums)}
    out["max_pair_error"], out["max_node_error"] = float(mx[0]), float(mx[1])
    return out


def graphcut_voxel_slabs(image, fg, bg, term="difference_exponential", sigma=None, spacing=False, nslabs=2, device=0,
                         connectivity=6, regional=None, **schedule):
    """Cut one volume as ``nslabs`` Z-slabs time-multiplexed on ONE GPU with the loopback transport.

    Exercises exactly the code path of the multi-GPU run (ghost layers, halo pack/unpack, the
    distributed schedule) where only one MI355X is available.  Returns (labels, flow, stats)."""
    image, fg, bg = np.asarray(image), np.asarray(fg), np.asarray(bg)
    slabs = [HipSlab(image.shape, r, nslabs, device=device, connectivity=connectivity) for r in range(nslabs)]
    for s in slabs:
        sl = slice(s.plane0, s.plane1)
        s.set_boundary(term, image[sl], sigma, spacing)
        if regional is not None:  # (probability map, alpha): regional_probability_map, energy_voxel.py:33-65
            s.set_regional(np.asarray(regional[0])[sl], regional[1])
        s.set_markers(fg[sl], bg[sl])
    ex = LoopbackExchange(slabs)
    if term.endswith("linear"):
        sync_image_range(slabs, ex)
    sync_boundary_table(slabs, ex)
    for s in slabs:
        s.build()
    st = solve_slabs(slabs, ex, **schedule)
    if not st.get("converged", 1):  # (solve_native hands back the stats of a run that exhausted max_outer instead of raising)
        for s in slabs:
            s.close()
        raise RuntimeError("graphcut_voxel_slabs: the slab schedule did not reach a maximum preflow within max_outer global relabels: %r" % (st,))
    parts = [s.finish() for s in slabs]
    labels = np.concatenate([p[0] for p in parts], axis=0)
    flow = float(sum(p[1] for p in parts))
    for s in slabs:
        s.close()
    return labels, flow, st
