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

This module holds the schedule (a mirror of ``medpy_amd/csrc/mgc_driver.inl`` with exchanges
and all-reduces added) and the two transports.  It is backend-agnostic: the CPU test tier runs
the same code over the host simulator, two processes over gloo or over a directory of files (tests/test_slab_*.py).
"""
import numpy as np

(OP_ABSORB_ALL, OP_FILL_INF, OP_ZERO_COUNT, OP_RELABEL_ALL, OP_RELABEL_LIST, OP_ACTIVATE, OP_DISCHARGE, OP_SUSPECT_PASS,
 OP_RESET_SUSPECT) = range(9)
CNT_CHANGED = 21  # MGC_CNT_CHANGED (mgc_common.h)
CNT_DEFERRED = 28  # MGC_CNT_DEFERRED: border tiles a full message left for the next exchange


class LoopbackExchange(object):
    """All slabs live in this process (one GPU time-multiplexed, or the host simulator)."""

    def __init__(self, slabs):
        self.slabs = list(slabs)
        self._bufs = {}
        self.on_device = False

    def _buf(self, key, nbytes):
        b = self._bufs.get(key)
        if b is None or b.size != nbytes:
            b = np.zeros(nbytes, dtype=np.uint8)
            self._bufs[key] = b
        return b

    def _raw(self, b):
        return b

    def exchange(self, kind, epoch, lst):
        packed = []
        for i in range(len(self.slabs) - 1):
            lo, hi = self.slabs[i], self.slabs[i + 1]
            nb = lo.halo_bytes(kind)
            up = self._buf((i, "up", kind), nb)
            dn = self._buf((i, "dn", kind), nb)
            lo.halo_pack(1, kind, self._raw(up), on_device=self.on_device)
            hi.halo_pack(0, kind, self._raw(dn), on_device=self.on_device)
            packed.append((up, dn))
        for i, (up, dn) in enumerate(packed):
            self.slabs[i + 1].halo_unpack(0, kind, self._raw(up), epoch, lst, on_device=self.on_device)
            self.slabs[i].halo_unpack(1, kind, self._raw(dn), epoch, lst, on_device=self.on_device)

    def allreduce_sum(self, values):
        """values: one number (or vector) per local slab -> global sum"""
        return np.sum(np.asarray(values, dtype=np.float64), axis=0)

    def allreduce_max(self, values):
        """values: one vector per local slab -> element-wise global maximum"""
        return np.max(np.asarray(values, dtype=np.float64), axis=0)

    def global_counts(self):
        """the 16 solver counters summed over every slab of the volume"""
        return np.sum([s.read_counts().astype(np.int64) for s in self.slabs], axis=0)


class StoreExchange(object):
    """One slab per process; the packed borders travel through host buffers and the out-of-band store
    (medpy_amd.rendezvous.FileStore: a file per message).  A DEVELOPMENT transport -- ranks may share one GPU, nothing here is
    fast -- that exercises the multi-process schedule where no second GPU (or no RCCL) is to be had: the CPU test tier over the
    host simulator and the reduced-size multi-rank run of bench.py.  The production transport is RcclExchange below."""

    def __init__(self, slab, store):
        self.store, self.slabs = store, [slab]
        self.rank, self.world = store.rank, store.world

    def exchange(self, kind, epoch, lst):
        slab = self.slabs[0]
        nb = slab.halo_bytes(kind)
        peers = [(side, peer) for side, peer in ((0, self.rank - 1), (1, self.rank + 1)) if 0 <= peer < self.world]
        for side, peer in peers:
            buf = np.zeros(nb, dtype=np.uint8)
            slab.halo_pack(side, kind, buf, on_device=False)
            self.store.send(peer, buf.tobytes())
        for side, peer in peers:
            buf = np.frombuffer(self.store.recv(peer, nb), dtype=np.uint8).copy()
            slab.halo_unpack(side, kind, buf, epoch, lst, on_device=False)

    def allreduce_sum(self, values):
        out = self.store.allreduce(np.sum(np.asarray(values, dtype=np.float64), axis=0), "sum")
        return out if out.size > 1 else float(out[0])

    def allreduce_max(self, values):
        return self.store.allreduce(np.max(np.asarray(values, dtype=np.float64), axis=0), "max")

    def global_counts(self):
        return np.asarray(self.allreduce_sum([self.slabs[0].read_counts().astype(np.float64)])).astype(np.int64)


class RcclExchange(object):
    """One slab per process, borders moved by the library itself: pack -> grouped ncclSend/ncclRecv with
    rank-1 / rank+1 -> unpack, stream-ordered in HBM (``mgc_halo_exchange``); counters summed with
    ncclAllReduce (``mgc_allreduce_counts``).  ``store`` (medpy_amd.rendezvous.FileStore, or anything with ``rank``, ``world``,
    ``broadcast(bytes, src, nbytes)`` and ``allreduce(array, op)``) is only the out-of-band channel that hands rank 0's
    128-byte RCCL id to the other ranks and sums a few host scalars at the end -- no PyTorch anywhere."""

    def __init__(self, slab, store):
        self.store = store
        self.slabs = [slab]
        self.rank, self.world = store.rank, store.world
        uid = slab.comm_unique_id() if self.rank == 0 else b""
        slab.comm_init(store.broadcast(uid, src=0, nbytes=128))

    native = True  # solve_slabs hands the whole schedule to the library (mgc_solve_slab)

    def exchange(self, kind, epoch, lst):
        self.slabs[0].exchange(kind, epoch, lst)

    def global_counts(self):
        return self.slabs[0].allreduce_counts()

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


def solve_slabs(slabs, ex, rounds_per_relabel=None, max_cycles=None, max_sweeps=None, max_outer=100000, check_rounds=4, relabel_batch=8,
                incremental_relabel=True, exchange_every=4):
    """Drives the local slabs to a maximum preflow.  Returns a stats dict (global numbers).

    Every loop decision that involves the other ranks is taken on globally summed counters, so all ranks run the same
    outer control flow.  Between two border exchanges of a global relabel every rank iterates its own slabs to a LOCAL
    fixpoint (labels only go down during a relabel, so stale ghost labels are upper bounds and the chaotic iteration
    still converges to the exact distances); the borders are exchanged after every round of the two colours (after every
    colour phase in the 26-neighbourhood, whose pushes land in the ghost tiles in place).  Later global
    relabels are incremental like the single-GPU driver's (mgc_driver.inl): the DIRTY / SUSPECT flags of the border
    tiles travel as halo kind 2 until the suspect closure is stable everywhere."""
    if getattr(ex, "native", False) and len(slabs) == 1 and hasattr(slabs[0], "solve_native"):
        # the library's own transport: the schedule below runs inside libmedpyhip (mgc_solve_slab), no call per kernel from here
        # None = the library's default for the slab's neighbourhood (rounds_per_relabel: 8 / 6)
        for name, value in (("rounds_per_relabel", rounds_per_relabel), ("max_cycles", max_cycles), ("max_sweeps", max_sweeps),
                            ("max_outer", max_outer), ("check_rounds", check_rounds), ("relabel_batch", relabel_batch),
                            ("incremental_relabel", int(bool(incremental_relabel))), ("relabel_exchange_every", exchange_every)):
            if value is not None:
                slabs[0].set_param(name, value)
        return slabs[0].solve_native()  # (converged == 0 in the stats when max_outer ran out, as the Python schedule reports it)
    if rounds_per_relabel is None:
        rounds_per_relabel = 6 if getattr(slabs[0], "ndir", 6) == 26 else 8
    relabel_batch = max(2, relabel_batch + (relabel_batch & 1))  # even: every rank keeps the same list parity
    # where the solver variant keeps its lists / counters (MgcLayout, mgc_driver.inl:51-62)
    if getattr(slabs[0], "ndir", 6) == 26:
        ncol, lmask, rl, c_act, c_dis, c_rel = 8, 15, 16, 18, 19, 20
        max_sweeps = max_sweeps or 3  # mgc_default_params(26)
        max_cycles = max_cycles or -1  # stored labels, mgc_default_params(26)
    else:
        ncol, lmask, rl, c_act, c_dis, c_rel = 2, 3, 4, 6, 8, 9
        max_sweeps = max_sweeps or 12  # mgc_default_params(6)
        max_cycles = max_cycles or 1
    phase, rep = 2 * (lmask + 1), 2
    for s in slabs:
        s.op(OP_ZERO_COUNT, c_dis)
        s.op(OP_ZERO_COUNT, c_rel)
    st = {"outer": 0, "relabel_passes": 0, "phases": 0, "exchanges": 0, "reductions": 0, "converged": 0}

    def exchange(kind, epoch, lst):
        ex.exchange(kind, epoch, lst)
        st["exchanges"] += 1

    def global_counts():
        st["reductions"] += 1
        return ex.global_counts()

    for s in slabs:
        s.op(OP_ZERO_COUNT, CNT_DEFERRED)
    for outer in range(max_outer):
        # ---- flow that a full border message left behind during the colour phases has to cross before the masks are read
        while outer > 0 and int(global_counts()[CNT_DEFERRED]) != 0:
            for s in slabs:
                s.op(OP_ZERO_COUNT, CNT_DEFERRED)
            exchange(1, phase - 1, 0)
            st["deferred_drains"] = st.get("deferred_drains", 0) + 1
        # ---- global relabel: tile BFS passes to a local fixpoint, border label exchange, until nothing moves anywhere
        for s in slabs:
            s.op(OP_ABSORB_ALL)
            s.op(OP_ZERO_COUNT, rl)
            s.op(OP_ZERO_COUNT, rl + 1)
        nxt = rl + ((rep + 1) & 1)
        if outer == 0 or not incremental_relabel:
            for s in slabs:
                s.op(OP_FILL_INF)
                s.op(OP_RELABEL_ALL, rep + 1, nxt)
        else:
            while True:  # which tiles may have lost the support of their labels (closure across the slab borders)
                for s in slabs:
                    s.op(OP_ZERO_COUNT, CNT_CHANGED)
                    for _b in range(8):
                        s.op(OP_SUSPECT_PASS)
                exchange(2, 0, 0)
                if global_counts()[CNT_CHANGED] == 0:
                    break
            for s in slabs:
                s.op(OP_RESET_SUSPECT, rep + 1, nxt)
        st["relabel_passes"] += 1
        rounds_done = 0
        while exchange_every > 0:
            # Border labels travel every `xk` passes, whether or not a slab has reached its local fixpoint: the label wave of the
            # relabel crosses a slab border at most xk passes after it reaches it (labels only go down during a relabel, so a ghost
            # label is an upper bound whenever it is read).  Every rank runs the same number of passes; counters are compared
            # every second exchange.  (mgc_solve_slab runs the same loop.)
            xk = exchange_every + (exchange_every & 1)
            for _b in range(xk):
                rep += 1
                cur, nxt = rl + (rep & 1), rl + ((rep + 1) & 1)
                for s in slabs:
                    s.op(OP_ZERO_COUNT, nxt)
                    s.op(OP_RELABEL_LIST, cur, rep + 1, nxt)
                st["relabel_passes"] += 1
            if rounds_done % 2 == 0:
                for s in slabs:
                    s.op(OP_ZERO_COUNT, CNT_DEFERRED)
            exchange(0, rep + 1, nxt)
            rounds_done += 1
            if rounds_done % 2 == 1:
                continue
            g = global_counts()
            if g[nxt] == 0 and g[CNT_DEFERRED] == 0:
                break
        while exchange_every <= 0:  # round 3's schedule: every slab to its local fixpoint between two exchanges
            while any(int(s.read_counts()[nxt]) != 0 for s in slabs):  # local read-back, no collective
                for _b in range(relabel_batch):
                    rep += 1
                    cur, nxt = rl + (rep & 1), rl + ((rep + 1) & 1)
                    for s in slabs:
                        s.op(OP_ZERO_COUNT, nxt)
                        s.op(OP_RELABEL_LIST, cur, rep + 1, nxt)
                    st["relabel_passes"] += 1
            for s in slabs:
                s.op(OP_ZERO_COUNT, CNT_DEFERRED)
            exchange(0, rep + 1, nxt)
            g = global_counts()
            if g[nxt] == 0 and g[CNT_DEFERRED] == 0:  # the exchange woke nobody anywhere and left nothing behind: global fixpoint
                break
        st["outer"] += 1

        # ---- who can still push towards the sink?
        phase += 2 * (lmask + 1)  # fresh stamps: anything queued before the relabel is void
        for s in slabs:
            for i in list(range(lmask + 1)) + [c_act]:
                s.op(OP_ZERO_COUNT, i)
            s.op(OP_ACTIVATE, phase)
        if global_counts()[c_act] == 0:
            st["converged"] = 1
            break

        # ---- colour phases, border (labels + outbox flow) exchanged after each
        for s in slabs:
            s.op(OP_ZERO_COUNT, CNT_DEFERRED)
        for r in range(rounds_per_relabel):
            for _c in range(ncol):
                lst = phase & lmask
                for s in slabs:
                    s.op(OP_DISCHARGE, lst, phase, max_cycles, max_sweeps)
                    s.op(OP_ZERO_COUNT, lst)
                if ncol != 2 or _c == 1:  # 6-neighbourhood: once per round of the two colours (mgc_halo_unpack_tile queues by colour)
                    exchange(1, phase, 0)
                st["phases"] += 1
                phase += 1
            if (r + 1) % check_rounds == 0 and r + 1 < rounds_per_relabel:
                c = global_counts()
                if int(np.sum(c[:lmask + 1])) + int(c[CNT_DEFERRED]) == 0:
                    break
    c = ex.global_counts()
    st["discharge_tiles"], st["relabel_tiles"] = int(c[c_dis]), int(c[c_rel])
    return st


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

    def solve_native(self):
        """the distributed schedule inside the library (mgc_solve_slab): needs comm_init() when the volume has several slabs"""
        st = self._lib.SlabStats()
        try:
            self._call("mgc_solve_slab", self._C.byref(st))
        except self._lib.MedpyHipError as err:
            if err.code != self._lib.ERR_NOT_CONVERGED:
                raise  # (the stats are filled before the library reports max_outer exhausted: converged stays 0)
        return st.as_dict()

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
