"""Graph facade of the voxel graph-cut path.

Mirrors reference medpy/graphcut/graph.py:267-596 (``GCGraph``) and the object it wraps,
``maxflow.GraphDouble`` (lib/maxflow/src/wrapper.cpp:59-89): same method names, argument
meaning and ``ValueError`` contract -- but instead of inserting one edge per Python call into
a CPU adjacency list, the facade records what the energy terms ask for and hands whole
arrays to the HIP library, which builds the residual lattice in HBM and solves it there.
"""
import ctypes as C
import enum

import numpy

from .. import _lib


class termtype(enum.IntEnum):
    """reference lib/maxflow/src/graph.h:57-61, exposed per class by wrapper.cpp:84-87."""
    SOURCE = 0
    SINK = 1


def _holds_whole_numbers(image, block=1 << 20):
    """every value of a float array is a whole number (NaN / inf: no) -- in blocks, leaving at the first block that holds
    anything else: a 512^3 float32 volume of noise is turned away after 4 MB, and no volume-sized temporary is made"""
    flat = image.reshape(-1) if image.flags.c_contiguous else image.ravel()
    for start in range(0, flat.size, block):
        part = flat[start:start + block]
        if not numpy.array_equal(part, numpy.floor(part)):   # (NaN != NaN, inf == floor(inf): caught by the range check below)
            return False
    return True


def image_table_facts(term, image):
    """what ``boundary_table`` needs to know about an image (or about one slab of it, medpy_amd/slab.py:sync_boundary_table):
    (holds whole numbers only, min, max) -- or None when the term has no table or the image no finite range"""
    import math
    if not (term.endswith("exponential") or term.endswith("power")):
        return None
    image = numpy.asarray(image)
    if image.size == 0 or image.dtype.kind not in "iuf":
        return None
    if image.dtype.kind == "f" and not _holds_whole_numbers(image):
        return False, 0.0, 0.0   # no table whatever the range: not worth two more passes over the volume (20 - 80 ms at 512^3)
    whole = True
    lo, hi = float(image.min()), float(image.max())
    if not (math.isfinite(lo) and math.isfinite(hi)):
        return None
    return whole, lo, hi


def boundary_table_for_range(term, sigma, lo, hi, limit=65536):
    """the table of ``boundary_table`` for an image known to hold whole numbers in [lo, hi]"""
    import math
    import sys
    if not (term.endswith("exponential") or term.endswith("power")) or sigma is None:
        return None
    top = max(abs(lo), abs(hi)) if term.startswith("maximum") else hi - lo
    if top + 1 > limit:
        return None
    x = numpy.arange(int(top) + 1, dtype=float)
    if term.endswith("exponential"):
        x = numpy.power(x, 2)
        x /= math.pow(sigma, 2)
        x *= -1
        x = numpy.exp(x)
    else:
        x = 1.0 / (x + 1)
        x = numpy.power(x, sigma)
    x[x <= 0] = sys.float_info.min
    return numpy.ascontiguousarray(x, dtype=numpy.float64)


def boundary_table(term, image, sigma, limit=65536):
    """The exponential / power boundary function of an INTEGER-VALUED image by table, or None.

    On such images (CT / MR data: uint8, uint16, int16; floats that hold whole numbers) the reference's term functions see
    only whole-number arguments d = |I_p - I_q| (difference terms) or max(|I_p|, |I_q|) (maximum terms), because
    ``__skeleton_base`` casts the image to float64 first (reference energy_voxel.py:633-634).  Evaluating the term ONCE per
    possible d with the very NumPy operations the reference applies to its arrays (energy_voxel.py:226-236 / 290-300:
    power(x, 2), /= pow(sigma, 2), *= -1, exp, floor at float_info.min; 444-452 / 506-513: 1 / (x + 1), power(x, sigma), floor)
    makes the n-link weights bit-identical to the reference's -- the device's own exp / pow (OCML) is up to 2 ulp away, which
    is enough to flip a tie.  The linear and division terms are IEEE-basic arithmetic and need no table."""
    if sigma is None:
        return None
    facts = image_table_facts(term, image)
    if facts is None or not facts[0]:
        return None
    return boundary_table_for_range(term, sigma, facts[1], facts[2], limit)


class VoxelGraph(object):
    """What ``graph_from_voxels`` returns: the stand-in for ``maxflow.GraphDouble``.

    Supports what callers of the reference use on the returned object
    (bin/medpy_graphcut_voxel.py:172-181, tests/graphcut_/energy_voxel.py:205-224):
    ``maxflow()``, ``what_segment(i)``, ``termtype``, ``get_edge``, ``get_node_num``,
    ``get_trcap`` -- plus bulk ``labels()`` so that nobody has to loop over voxels in Python.
    """

    termtype = termtype

    def __init__(self, shape, device=0, connectivity=None):
        lib = _lib.load()
        if _lib.device_count() < 1:
            raise _lib.MedpyHipError(_lib.ERR_NO_DEVICE, "no HIP device visible; medpy_amd has no CPU fallback")
        self._shape = tuple(int(s) for s in shape)
        nd = len(self._shape)
        if nd < 1 or nd > 3:
            raise NotImplementedError("medpy_amd: %d-D lattices are not implemented (1-D..3-D are)" % nd)
        shp = (C.c_int64 * nd)(*self._shape)
        h = C.c_void_p()
        rc = lib.mgc_create(nd, shp, connectivity or 2 * nd, int(device), C.byref(h))
        self._h = h if h.value else None
        if rc != _lib.OK:
            msg = (lib.mgc_last_error(self._h) or b"").decode()
            self.close()
            raise _lib.MedpyHipError(rc, msg)
        self._nodes = int(numpy.prod(self._shape))
        self._labels = None
        _lib.apply_env_params(self._h)

    # -- life cycle
    def close(self):
        if getattr(self, "_h", None):
            _lib.load().mgc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        _lib.check(self._h, getattr(_lib.load(), name)(self._h, *args))

    # -- inputs (called by GCGraph)
    def _set_boundary(self, term, image, sigma, spacing):
        image = numpy.ascontiguousarray(image)
        if image.dtype == numpy.bool_:
            image = image.astype(numpy.uint8)
        if image.dtype == numpy.float16:
            image = image.astype(numpy.float32)
        if image.dtype not in _lib.DTYPE_IDS:
            image = image.astype(numpy.float64)
        sp = None
        if spacing:
            sp = (C.c_double * len(self._shape))(*[float(s) for s in spacing])
        self._call("mgc_set_boundary", _lib.TERM_IDS[term], _lib.ptr(image), _lib.DTYPE_IDS[image.dtype],
                   float(sigma) if sigma is not None else 0.0, sp)
        table = boundary_table(term, image, sigma)
        if table is not None:
            self._call("mgc_set_boundary_lut", _lib.ptr(table), table.size)

    def _set_regional(self, prob, alpha):
        prob = numpy.asarray(prob)
        if prob.dtype not in (numpy.float32, numpy.float64):
            prob = prob.astype(numpy.float64)  # NumPy promotes (non-float array * python float) to float64
        prob = numpy.ascontiguousarray(prob)
        self._call("mgc_set_regional_probability", _lib.ptr(prob), _lib.DTYPE_IDS[prob.dtype], float(alpha))

    def _set_markers(self, fg, bg):
        def as_bytes(m):   # a C-contiguous bool array IS the byte array the library reads: no 1-byte-per-voxel copy (0.1 s at 512^3)
            if m is None:
                return None
            m = numpy.asarray(m)
            if m.dtype in (numpy.bool_, numpy.uint8, numpy.int8) and m.flags.c_contiguous:
                return m.view(numpy.uint8)   # (the library reads "non-zero")
            return numpy.ascontiguousarray(m, dtype=numpy.bool_).view(numpy.uint8)
        fg8, bg8 = as_bytes(fg), as_bytes(bg)
        self._call("mgc_set_markers", None if fg8 is None else _lib.ptr(fg8), None if bg8 is None else _lib.ptr(bg8))

    def _add_edges(self, i, j, cap, rev):
        i = numpy.ascontiguousarray(i, dtype=numpy.int64)
        j = numpy.ascontiguousarray(j, dtype=numpy.int64)
        cap = numpy.ascontiguousarray(cap, dtype=numpy.float64)
        rev = numpy.ascontiguousarray(rev, dtype=numpy.float64)
        self._call("mgc_add_edges", i.size, _lib.ptr(i), _lib.ptr(j), _lib.ptr(cap), _lib.ptr(rev))

    def _set_tweights_merged(self, tr, flow_const):
        tr = numpy.ascontiguousarray(tr, dtype=numpy.float64)
        self._call("mgc_set_tweights_merged", _lib.ptr(tr), float(flow_const))

    def _build(self):
        self._call("mgc_build")
        self._labels = None

    def set_param(self, name, value):
        self._call("mgc_set_param", name.encode(), int(value))

    def validate(self):
        """invariants of the maximum preflow in HBM (mgc_validate): dict of violation counts (all zero for a correct
        solve), the two conservation errors, the flow into the sink and the capacity of the cut"""
        v = _lib.Validation()
        self._call("mgc_validate", C.byref(v))
        return v.as_dict()

    # -- GraphDouble surface
    def maxflow(self):
        """GraphDouble.maxflow(), reference maxflow.cpp:472-604."""
        flow = C.c_double(0.0)
        self._call("mgc_maxflow", C.byref(flow))
        return flow.value

    def labels(self):
        """All voxels at once: bool array of the marker shape, False where what_segment == SINK
        (the loop of bin/medpy_graphcut_voxel.py:177-182)."""
        if self._labels is None:
            out = numpy.empty(self._nodes, dtype=numpy.uint8)
            self._call("mgc_labels", _lib.ptr(out))
            self._labels = out.view(numpy.bool_).reshape(self._shape)   # (the library writes 0 / 1: the bytes ARE the bool array, no second pass over the volume)
        return self._labels

    def what_segment(self, i):
        """Graph::what_segment, reference graph.h:561-571."""
        seg = C.c_int(0)
        self._call("mgc_what_segment", int(i), C.byref(seg))
        return termtype(seg.value)

    def get_edge(self, i, j):
        out = C.c_double(0.0)
        self._call("mgc_get_edge", int(i), int(j), C.byref(out))
        return out.value

    def get_node_num(self):
        return self._nodes

    def get_trcap(self, i):
        return float(self.tweights().ravel()[int(i)])

    # -- energy read-back (parity tests)
    def nweights_offset(self, offset):
        """weights of the arcs (p, p + offset), NaN where p + offset is outside the volume"""
        off = (C.c_int * len(self._shape))(*[int(v) for v in offset])
        out = numpy.empty(self._shape, dtype=numpy.float64)
        self._call("mgc_get_nweights_offset", off, _lib.ptr(out))
        return out

    def nweights(self, axis):
        shp = list(self._shape)
        shp[axis] -= 1
        out = numpy.empty(shp, dtype=numpy.float64)
        if out.size:
            self._call("mgc_get_nweights", int(axis), _lib.ptr(out))
        return out

    def tweights(self):
        out = numpy.empty(self._shape, dtype=numpy.float64)
        self._call("mgc_get_tweights", _lib.ptr(out))
        return out

    def profile(self):
        out = numpy.zeros(16, dtype=numpy.uint64)
        self._call("mgc_get_profile", _lib.ptr(out))
        names = ("load", "labels", "sweep", "store", "votes", "faceflags", "s6", "s7")
        return {n: {"cycles": int(out[i]), "count": int(out[i + 8])} for i, n in enumerate(names)}

    def stats(self):
        st = _lib.Stats()
        self._call("mgc_get_stats", C.byref(st))
        return st.as_dict()


class SparseGraph(object):
    """``maxflow.GraphDouble`` (reference lib/maxflow/src/wrapper.cpp:59-89) for ARBITRARY graphs, solved in HBM by the
    library's sparse-graph solver (C ABI ``msg_*``).  Returned by ``graph_from_labels``, by ``graph_from_voxels`` for
    images of more than three dimensions and by ``GCGraph.get_graph()`` for graphs that plug-in terms assemble edge by
    edge.  Besides the facade hooks (``_add_*``) it takes the raw GraphDouble calls ``add_node``, ``add_edge``,
    ``sum_edge``, ``add_tweights``, ``get_edge`` and ``reset`` with the reference's semantics (graph.h:428-498,
    graph.cpp:46-60): ``add_edge`` creates a PARALLEL arc pair per call (the flow sees the summed capacity; ``get_edge``
    reports the arc the reference's list walk meets first, i.e. the pair added last), ``sum_edge`` adds to that arc.
    Edges are buffered and uploaded in batches.

    ``_cast`` is the capacity type of the instance (``GraphDouble``: float64; the subclasses ``GraphFloat`` / ``GraphInt``
    quantise every capacity, every running sum and the returned flow to float32 / int, instances.inc:12-15)."""

    termtype = termtype
    _captype = "double"

    @staticmethod
    def _cast(v):
        """a capacity handed in by the caller, in the graph's capacity type"""
        return float(v)

    @classmethod
    def _out(cls, v):
        """a value read back from the device, in the graph's capacity type"""
        return cls._cast(v)

    def __init__(self, nodes, edges=0, device=0):
        lib = _lib.load()
        if _lib.device_count() < 1:
            raise _lib.MedpyHipError(_lib.ERR_NO_DEVICE, "no HIP device visible; medpy_amd has no CPU fallback")
        self._nodes = max(int(nodes), 1)
        self._device = int(device)
        self._h = None
        self._raw = False   # driven through the raw GraphDouble calls (add_node / add_edge / sum_edge / reset) rather than the facade hooks
        self._open()

    def _open(self):
        """a fresh, empty solver graph in HBM (``__init__`` and ``reset``)"""
        lib = _lib.load()
        self.close()
        h = C.c_void_p()
        rc = lib.msg_create(self._nodes, self._device, C.byref(h))
        self._h = h if h.value else None
        if rc != _lib.OK:
            msg = (lib.msg_last_error(self._h) or b"").decode()
            self.close()
            raise _lib.MedpyHipError(rc, msg)
        self._labels = None
        self._pending = ([], [], [], [])
        self._tr = None
        self._flow_const = self._cast(0)
        self._declared = 0
        self._first_arc = {}   # (i, j) -> capacity of the arc get_arc(i, j) meets first (raw add_edge / sum_edge calls only)
        self._raw_pairs = set()
        self._host_arcs = {}   # GraphFloat / GraphInt: (i, j) -> [capacity of the front arc, sum of the parallel arcs behind it]
        self._host_sent = {}   # ... (i, j) with i < j -> (capacity, reverse capacity) the device holds for the pair so far

    def reset(self):
        """Graph::reset, reference graph.cpp:46-60 (wrapper.cpp:68): back to the state just after construction -- no nodes
        declared, no arcs, no t-links, flow 0."""
        self._raw = True
        self._open()

    def close(self):
        if getattr(self, "_h", None):
            _lib.load().msg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        _lib.check_sparse(self._h, getattr(_lib.load(), name)(self._h, *args))

    @staticmethod
    def _image(image):
        image = numpy.ascontiguousarray(image)
        if image.dtype == numpy.bool_:
            image = image.astype(numpy.uint8)
        if image.dtype == numpy.float16:
            image = image.astype(numpy.float32)
        if image.dtype not in _lib.DTYPE_IDS:
            image = image.astype(numpy.float64)
        return image

    # -- inputs
    def _flush(self):
        i, j, cap, rev = self._pending
        if i:
            self._pending = ([], [], [], [])
            self._add_edges(i, j, cap, rev)

    def _add_edges(self, i, j, cap, rev):
        i = numpy.ascontiguousarray(i, dtype=numpy.int64)
        j = numpy.ascontiguousarray(j, dtype=numpy.int64)
        cap = numpy.ascontiguousarray(cap, dtype=numpy.float64)
        rev = numpy.ascontiguousarray(rev, dtype=numpy.float64)
        self._call("msg_add_edges", i.size, _lib.ptr(i), _lib.ptr(j), _lib.ptr(cap), _lib.ptr(rev))
        self._labels = None

    def _add_lattice_edges(self, term, image, sigma, spacing):
        self._flush()
        image = self._image(image)
        shp = (C.c_int64 * image.ndim)(*image.shape)
        sp = (C.c_double * image.ndim)(*[float(v) for v in spacing]) if spacing else None
        self._call("msg_add_lattice_edges", _lib.TERM_IDS[term], image.ndim, shp, _lib.ptr(image), _lib.DTYPE_IDS[image.dtype],
                   float(sigma) if sigma is not None else 0.0, sp)
        self._labels = None

    def _add_label_edges(self, term, label_image, image, param=0.0):
        self._flush()
        lab = numpy.ascontiguousarray(label_image, dtype=numpy.int64)
        image = self._image(numpy.asarray(image))
        if image.shape != lab.shape:
            raise ValueError("label image {} and image {} differ in shape".format(lab.shape, image.shape))
        shp = (C.c_int64 * lab.ndim)(*lab.shape)
        self._call("msg_add_label_edges", _lib.LABEL_TERM_IDS[term], lab.ndim, shp, _lib.ptr(lab), _lib.ptr(image),
                   _lib.DTYPE_IDS[image.dtype], float(param))
        self._labels = None

    def _set_tweights_merged(self, tr, flow_const):
        self._tr = numpy.array(tr, dtype=numpy.float64)
        self._flow_const = float(flow_const)
        self._labels = None

    def set_param(self, name, value):
        self._call("msg_set_param", name.encode(), int(value))

    # -- raw GraphDouble calls (graph.h:388-480)
    def add_node(self, num=1):
        """Graph::add_node, reference graph.h:388-413: declares ``num`` more nodes, returns the id of the first.  Going beyond
        the constructor's node count grows the graph (the reference reallocates, graph.cpp:62-85): a larger graph is
        created in HBM and what was uploaded so far moves over."""
        first = self._declared
        self._raw = True
        self._declared += int(num)
        if self._declared > self._nodes:
            self._grow(self._declared)
        return first

    def _grow(self, nodes):
        self._flush()
        tail, head, cap = self.arcs()
        lib = _lib.load()
        old = self._h
        h = C.c_void_p()
        rc = lib.msg_create(int(nodes), self._device, C.byref(h))
        if rc != _lib.OK:
            msg = (lib.msg_last_error(h if h.value else None) or b"").decode()
            if h.value:
                lib.msg_destroy(h)
            raise _lib.MedpyHipError(rc, msg)
        self._h, self._nodes = h, int(nodes)
        lib.msg_destroy(old)
        if tail.size:
            self._add_edges(tail, head, cap, numpy.zeros(cap.size))
        if self._tr is not None:
            self._tr = numpy.concatenate([self._tr, numpy.zeros(self._nodes - self._tr.size)])
        self._labels = None

    def _queue(self, i, j, cap, rev_cap):
        p = self._pending
        p[0].append(i); p[1].append(j); p[2].append(float(cap)); p[3].append(float(rev_cap))
        self._labels = None
        if len(p[0]) >= 1 << 20:
            self._flush()

    def _check_ids(self, i, j):
        n = max(self._declared, self._nodes) if self._declared else self._nodes
        if not (0 <= i < n and 0 <= j < n) or i == j:   # the reference asserts (graph.h:430-432); here an exception
            raise ValueError("edge ({}, {}): node ids must differ and lie in [0, {})".format(i, j, n))

    def add_edge(self, i, j, cap, rev_cap):
        """Graph::add_edge, reference graph.h:428-454: a NEW pair of arcs i->j / j->i per call, prepended to both adjacency
        lists.  Parallel arcs carry the flow of one arc with the summed capacity (that is what goes to the device);
        ``get_edge`` afterwards sees the pair added LAST, as the reference's ``get_arc`` walk does (graph.h:500-509)."""
        i, j, cap, rev_cap = int(i), int(j), self._cast(cap), self._cast(rev_cap)
        self._check_ids(i, j)
        self._raw = True
        if self._captype != "double":
            for key, c in (((i, j), cap), ((j, i), rev_cap)):
                front = self._host_arcs.get(key)
                self._host_arcs[key] = [c, 0.0] if front is None else [c, front[1] + front[0]]
            self._labels = None
            return
        if (i, j) in self._raw_pairs or (j, i) in self._raw_pairs:
            self._first_arc[(i, j)], self._first_arc[(j, i)] = cap, rev_cap   # a parallel pair now hides the older ones
        self._raw_pairs.add((i, j))
        self._queue(i, j, cap, rev_cap)

    def sum_edge(self, i, j, cap, rev_cap):
        """Graph::sum_edge, reference graph.h:457-480: adds to the arc ``get_arc(i, j)`` finds (and to its sister), or
        creates the pair."""
        i, j, cap, rev_cap = int(i), int(j), self._cast(cap), self._cast(rev_cap)
        self._check_ids(i, j)
        if self._captype != "double":   # float32 / int arcs: the running sum is rounded per call, in the arc's type (host side)
            if (i, j) not in self._host_arcs:
                return self.add_edge(i, j, cap, rev_cap)
            a, r = self._host_arcs[(i, j)], self._host_arcs[(j, i)]
            a[0], r[0] = self._cast(a[0] + cap), self._cast(r[0] + rev_cap)
            self._labels = None
            return
        if (i, j) in self._first_arc:   # parallel pairs exist: the one in front takes the sum
            self._first_arc[(i, j)] = self._first_arc[(i, j)] + cap
            self._first_arc[(j, i)] = self._first_arc[(j, i)] + rev_cap
        self._raw_pairs.add((i, j))
        self._queue(i, j, cap, rev_cap)

    def add_tweights(self, i, cap_source, cap_sink):
        if self._tr is None:
            self._tr = numpy.zeros(self._nodes, dtype=numpy.float64)
        cs, ck = self._cast(cap_source), self._cast(cap_sink)
        delta = self._cast(self._tr[i])
        if delta > 0:
            cs = self._cast(cs + delta)
        else:
            ck = self._cast(ck - delta)
        self._flow_const = self._cast(self._flow_const + (cs if cs < ck else ck))
        self._tr[i] = self._cast(cs - ck)
        self._labels = None

    # -- GraphDouble surface
    def maxflow(self):
        """GraphDouble.maxflow(), reference maxflow.cpp:472-604"""
        self._flush()
        self._send_host_arcs()
        if self._tr is not None:
            tr = numpy.ascontiguousarray(self._tr, dtype=numpy.float64)
            self._call("msg_set_tweights_merged", _lib.ptr(tr), float(self._flow_const))
        flow = C.c_double(0.0)
        self._call("msg_maxflow", C.byref(flow))
        self._labels = None
        return self._out(flow.value)

    def _send_host_arcs(self):
        """GraphFloat / GraphInt: the arcs accumulated on the host go to the device, parallel arcs summed (exactly: float32
        values and integers add without rounding in float64).  The reference's Graph<int> / Graph<float> take ``add_edge`` /
        ``sum_edge`` at any time (graph.h:428-480), also between two ``maxflow()`` calls: what goes down here is, per node pair,
        the DIFFERENCE between the pair's capacities now and what the device was last told -- exact in float64 for the same
        reason the sums are -- so edges added after a first solve are part of the next one."""
        if not self._host_arcs:
            return
        ii, jj, cc, rr = [], [], [], []
        for k, a in self._host_arcs.items():
            if k[0] >= k[1]:
                continue
            b = self._host_arcs[(k[1], k[0])]
            now = (float(a[0]) + float(a[1]), float(b[0]) + float(b[1]))
            was = self._host_sent.get(k, (0.0, 0.0))
            if now != was:
                ii.append(k[0]); jj.append(k[1]); cc.append(now[0] - was[0]); rr.append(now[1] - was[1])
                self._host_sent[k] = now
        if ii:
            self._add_edges(numpy.array(ii, dtype=numpy.int64), numpy.array(jj, dtype=numpy.int64),
                            numpy.array(cc, dtype=numpy.float64), numpy.array(rr, dtype=numpy.float64))

    def labels(self):
        """every node at once: bool array, False where what_segment == SINK"""
        if self._labels is None:
            out = numpy.empty(self._nodes, dtype=numpy.uint8)
            self._call("msg_labels", _lib.ptr(out))
            self._labels = out.astype(numpy.bool_)
        return self._labels

    def what_segment(self, i):
        seg = C.c_int(0)
        self._call("msg_what_segment", int(i), C.byref(seg))
        return termtype(seg.value)

    def get_edge(self, i, j):
        """Graph::get_edge, reference graph.h:482-498: the capacity of the first arc i->j of i's adjacency list (the pair
        added last when ``add_edge`` created parallel ones), 0 when there is none"""
        i, j = int(i), int(j)
        if self._captype != "double":
            a = self._host_arcs.get((i, j))
            return self._cast(0) if a is None else a[0]
        if (i, j) in self._first_arc:
            return self._first_arc[(i, j)]
        self._flush()
        out = C.c_double(0.0)
        self._call("msg_get_edge", i, j, C.byref(out))
        return self._out(out.value)

    def get_trcap(self, i):
        return self._cast(0) if self._tr is None else self._cast(self._tr[int(i)])

    def get_node_num(self):
        """nodes declared with ``add_node`` so far (graph.h:413), or the constructor's count when the facade filled the graph"""
        return self._declared if self._raw else self._nodes

    def get_arc_num(self):
        self._flush()
        self._send_host_arcs()
        n = C.c_int64(0)
        self._call("msg_get_counts", None, None, C.byref(n))
        return n.value

    def arcs(self):
        """every distinct arc as built: (tail, head, capacity) arrays sorted by (tail, head)"""
        n = self.get_arc_num()
        tail, head, cap = numpy.zeros(n, numpy.int32), numpy.zeros(n, numpy.int32), numpy.zeros(n, numpy.float64)
        if n:
            self._call("msg_get_arcs", _lib.ptr(tail), _lib.ptr(head), _lib.ptr(cap))
        return tail, head, cap

    def tweights(self):
        return numpy.zeros(self._nodes) if self._tr is None else numpy.array(self._tr)

    def stats(self):
        st = _lib.SparseStats()
        self._call("msg_get_stats", C.byref(st))
        return st.as_dict()


class GraphFloat(SparseGraph):
    """``maxflow.GraphFloat`` = ``Graph<float,float,float>`` (reference instances.inc:14, wrapper.cpp:27-57): capacities,
    t-links, their running sums and the returned flow are float32 values.  The solve itself runs in the library's float64
    arithmetic on those float32-valued capacities (sums of float32 numbers are exact in float64 far beyond any graph's
    size), so the cut is the exact minimum cut of the float32 graph; the reference's float32 augmentations can differ
    from it only where its own rounding decides a tie."""

    _captype = "float"

    @staticmethod
    def _cast(v):
        return float(numpy.float32(v))


class GraphInt(SparseGraph):
    """``maxflow.GraphInt`` = ``Graph<int,int,int>`` (reference instances.inc:12, wrapper.cpp:93-134): integer capacities
    and flow.  Whole numbers below 2**53 are exact in float64, so the device solve is exact integer arithmetic; like the
    Boost.Python binding, a capacity that is not an integer is a ``TypeError`` (there is no implicit float -> int
    conversion at that boundary)."""

    _captype = "int"

    @staticmethod
    def _cast(v):
        if isinstance(v, (bool, numpy.bool_)):
            return int(v)
        if isinstance(v, (int, numpy.integer)):
            return int(v)
        if isinstance(v, numpy.floating) and float(v).is_integer():
            return int(v)   # (the host copy of the t-links is a float64 array that holds whole numbers)
        raise TypeError("GraphInt: capacity %r is not an integer" % (v,))

    @staticmethod
    def _out(v):
        return int(round(v))


def region_sums(label_image, values, nregions, device=0):
    """per-region sums of ``values`` for labels 1..nregions, computed in HBM (``msg_region_sums``); float32 maps keep a
    float32 accumulator like numpy.sum does (reference energy_label.py:394-397).  Returns (sums, counts)."""
    lab = numpy.ascontiguousarray(label_image, dtype=numpy.int64)
    values = SparseGraph._image(numpy.asarray(values))
    if values.shape != lab.shape:
        raise ValueError("label image {} and map {} differ in shape".format(lab.shape, values.shape))
    sums = numpy.zeros(int(nregions), dtype=numpy.float64)
    counts = numpy.zeros(int(nregions), dtype=numpy.int64)
    rc = _lib.load().msg_region_sums(int(device), lab.size, _lib.ptr(lab), _lib.ptr(values), _lib.DTYPE_IDS[values.dtype],
                                     int(values.dtype == numpy.float32), int(nregions), _lib.ptr(sums), _lib.ptr(counts))
    _lib.check_sparse(None, rc)
    return sums, counts


class EmbeddedLatticeGraph(object):
    """`nodes` graph nodes of which the first prod(lattice_shape) form a voxel lattice (the boundary image had another
    shape than the markers, see GCGraph.record_boundary); the other nodes carry only their marker t-links.  Same surface
    as VoxelGraph; the lattice part is solved on the GPU, the isolated nodes are read out by the sign of their t-link
    exactly as BK would (tr_cap < 0: sink tree root -> SINK; otherwise SOURCE, graph.h:561-571)."""

    termtype = termtype

    def __init__(self, nodes, lattice_shape, boundary, fg, bg, device=0, connectivity=None):
        self._nodes = int(nodes)
        n = int(numpy.prod(lattice_shape))
        fg = numpy.zeros(self._nodes, numpy.uint8) if fg is None else numpy.asarray(fg, dtype=numpy.uint8).ravel()
        bg = numpy.zeros(self._nodes, numpy.uint8) if bg is None else numpy.asarray(bg, dtype=numpy.uint8).ravel()
        self._inner = VoxelGraph(lattice_shape, device=device, connectivity=connectivity)
        self._inner._set_boundary(*boundary)
        self._inner._set_markers(fg[:n].reshape(lattice_shape), bg[:n].reshape(lattice_shape))
        self._inner._build()
        tr = fg[n:].astype(numpy.float64) * GCGraph.MAX - bg[n:].astype(numpy.float64) * GCGraph.MAX
        self._tail_labels = ~(tr < 0)
        self._tail_flow = float(numpy.sum(numpy.minimum(fg[n:], bg[n:]).astype(numpy.float64)) * GCGraph.MAX)  # graph.h:423
        self._n = n

    def maxflow(self):
        return self._inner.maxflow() + self._tail_flow

    def labels(self):
        return numpy.concatenate([self._inner.labels().ravel(), self._tail_labels])

    def what_segment(self, i):
        i = int(i)
        if i < self._n:
            return self._inner.what_segment(i)
        return termtype.SOURCE if self._tail_labels[i - self._n] else termtype.SINK

    def get_node_num(self):
        return self._nodes

    def get_edge(self, i, j):
        return self._inner.get_edge(i, j) if (i < self._n and j < self._n) else 0.0

    def stats(self):
        return self._inner.stats()


class Graph(object):
    """Host-side description of a small graph-cut problem, the input format of ``graph_to_dimacs``.

    Same public surface as the reference's container (medpy/graphcut/graph.py:31-264; what write.py:29-76 reads):
    nodes are numbered 1..n, ``set_nweights`` takes ``{(a, b): (w_ab, w_ba)}``, ``add_tweights`` takes
    ``{node: (w_source, w_sink)}``, marker nodes get a terminal weight of ``MAX``.  Holds no device state."""

    MAX = 65535

    def __init__(self):
        self._count = 0
        self._markers = {"source": [], "sink": []}
        self._edges = {}      # (a, b) -> (w_ab, w_ba), insertion order = output order
        self._terminal = {}   # node -> (w_source, w_sink), insertion order = output order

    # ---- building ----
    def set_nodes(self, nodes):
        self._count = int(nodes)

    def _mark(self, side, nodes):
        self._markers[side] = list(nodes)
        weight = (self.MAX, 0) if side == "source" else (0, self.MAX)
        self._terminal.update(dict.fromkeys(self._markers[side], weight))

    def set_source_nodes(self, source_nodes):
        self._mark("source", source_nodes)

    def set_sink_nodes(self, sink_nodes):
        self._mark("sink", sink_nodes)

    def set_nweights(self, nweights):
        self._edges = nweights

    def add_tweights(self, tweights):
        self._terminal.update(tweights)

    # ---- reading ----
    def get_node_count(self):
        return self._count

    def get_nodes(self):
        return list(range(1, self._count + 1))

    def get_source_nodes(self):
        return self._markers["source"]

    def get_sink_nodes(self):
        return self._markers["sink"]

    def get_edges(self):
        return list(self._edges)

    def get_nweights(self):
        return self._edges

    def get_tweights(self):
        return self._terminal

    def inconsistent(self):
        """``False`` for a well-formed graph, otherwise one message per defect, in the reference's wording and order
        (reference graph.py:227-264): a node id above the node count in the t-weights, the markers or an edge, and every
        edge whose reverse is stored as an edge of its own (the weights of both directions belong into ONE entry)."""
        def unknown(node):
            return not node <= self._count

        messages = ["Node {} in t-weights but not in nodes.".format(n) for n in self._terminal if unknown(n)]
        messages += ["Node {} in s-nodes but not in nodes.".format(n) for n in self._markers["source"] if unknown(n)]
        messages += ["Node {} in t-nodes but not in nodes.".format(n) for n in self._markers["sink"] if unknown(n)]
        for edge in self._edges:
            messages += ["Node {} in edge {} but not in nodes.".format(n, edge) for n in edge if unknown(n)]
            if (edge[1], edge[0]) in self._edges:
                messages.append("The reversed edges of {} is also in the n-weights.".format(edge))
        return messages or False


class GCGraph(object):
    """Validating facade handed to the energy terms; reference graph.py:267-596.

    ``shape`` (not in the reference signature) tells the facade which voxel lattice the node
    ids refer to; ``graph_from_voxels`` supplies it.  Built-in energy terms record themselves
    (``record_boundary`` / ``record_regional``); foreign callables may still drive
    ``set_nweight`` / ``set_tweight*`` -- those calls are validated exactly like the reference
    does, accumulated, and uploaded in one batch.
    """

    __INT_16_BIT = 32767
    __UINT_16_BIT = 65535
    MAX = __UINT_16_BIT
    """The maximum value a terminal weight can take."""

    def __init__(self, nodes, edges, shape=None, device=0, connectivity=None):
        self.__connectivity = connectivity
        self.__nodes = int(nodes)
        self.__edges = int(edges)
        self.__general = shape is None or len(tuple(shape)) > 3  # not a 1-D..3-D voxel lattice: sparse-graph solver
        self.__label_terms = []
        self.__shape = tuple(shape) if shape is not None else (int(nodes),)
        if int(numpy.prod(self.__shape)) != self.__nodes:
            raise ValueError("shape {} does not hold {} nodes".format(self.__shape, nodes))
        self.__device = device
        self.__boundary = None
        self.__regional = None
        self.__fg = None
        self.__bg = None
        self.__edge_i, self.__edge_j, self.__edge_w, self.__edge_r = [], [], [], []
        self.__tr = None  # merged explicit t-links (graph.h:416-425 applied call by call)
        self.__flow_const = 0.0
        self.__graph = None
        self.__lattice_shape = None  # set when the boundary image has another shape than the markers

    # -- fast path hooks used by medpy_amd.graphcut.energy_voxel
    def record_boundary(self, term, image, sigma, spacing):
        image = numpy.asarray(image)
        if image.shape != self.__shape:
            # The reference numbers the n-link endpoints by the IMAGE shape (energy_voxel.py:650-664), whatever the
            # marker shape is (its own tests do that: tests/graphcut_/energy_voxel.py:162-179, 4x4 markers, 3x3 image).
            # The edges then form a lattice of the image shape over node ids 0..image.size-1; the remaining nodes are
            # isolated.  Too many ids -> the same ValueError GCGraph.set_nweight raises (graph.py:418-425).
            if image.size > self.__nodes:
                raise ValueError("Invalid node id (node_to) of {}. Valid values are 0 to {}.".format(image.size - 1, self.__nodes - 1))
            if self.__regional is not None or self.__tr is not None or self.__edge_i:
                raise NotImplementedError("medpy_amd: a boundary image of another shape cannot be combined with other terms")
            self.__lattice_shape = image.shape
        if self.__boundary is not None:
            raise NotImplementedError("medpy_amd: only one built-in boundary term per graph")
        self.__boundary = (term, image, sigma, spacing)

    def record_label_boundary(self, term, label_image, image, param=0.0):
        """fast path of the region terms (medpy_amd.graphcut.energy_label): the RAG edges are generated in HBM"""
        self.__general = True
        self.__label_terms.append((term, numpy.asarray(label_image), numpy.asarray(image), float(param)))

    def merge_tweights(self, nodes, weights_source, weights_sink):
        """vectorised ``set_tweight`` for DISTINCT node ids, Graph::add_tweights call by call (graph.h:416-425)"""
        nodes = numpy.asarray(nodes, dtype=numpy.int64)
        if nodes.size == 0:
            return
        if nodes.max() >= self.__nodes or nodes.min() < 0:
            raise ValueError("Invalid node id of {} or {}. Valid values are 0 to {}.".format(nodes.max(), nodes.min(), self.__nodes - 1))
        if self.__tr is None:
            self.__tr = numpy.zeros(self.__nodes, dtype=numpy.float64)
        cs = numpy.array(weights_source, dtype=numpy.float64)
        ck = numpy.array(weights_sink, dtype=numpy.float64)
        delta = self.__tr[nodes]
        cs = cs + numpy.where(delta > 0, delta, 0.0)
        ck = ck - numpy.where(delta > 0, 0.0, delta)
        self.__flow_const = float(numpy.cumsum(numpy.concatenate([[self.__flow_const], numpy.minimum(cs, ck)]))[-1])  # in call order
        self.__tr[nodes] = cs - ck

    def record_regional(self, probability_map, alpha):
        pm = numpy.asarray(probability_map)
        if pm.size != self.__nodes:
            raise ValueError("probability map holds {} values for {} nodes".format(pm.size, self.__nodes))
        self.__regional = (pm.reshape(self.__shape), alpha)

    # -- reference API
    def record_markers(self, fg_mask, bg_mask):
        """fast path of ``graph_from_voxels``: the marker MASKS (bool arrays of the volume's shape) instead of id lists --
        what ``set_source_nodes(flatnonzero(fg))`` / ``set_sink_nodes(flatnonzero(bg))`` would leave, without the lists"""
        for mask, side in ((fg_mask, "fg"), (bg_mask, "bg")):
            mask = numpy.ascontiguousarray(mask, dtype=numpy.bool_)
            if mask.size != self.__nodes:
                raise ValueError("marker mask of {} voxels on a graph of {} nodes".format(mask.size, self.__nodes))
            if not mask.any():
                continue
            flat = mask.reshape(-1).view(numpy.uint8)
            have = self.__fg if side == "fg" else self.__bg
            if have is not None:   # ids were wired before (a plug-in called set_*_nodes): repeated ids accumulate, the explicit path keeps that
                (self.set_source_nodes if side == "fg" else self.set_sink_nodes)(numpy.flatnonzero(flat))
            elif side == "fg":
                self.__fg = flat
            else:
                self.__bg = flat

    def set_source_nodes(self, source_nodes):
        source_nodes = numpy.asarray(source_nodes)
        if source_nodes.size == 0:
            raise ValueError("max() arg is an empty sequence")  # the reference's max([]) (graph.py:334)
        if source_nodes.max() >= self.__nodes or source_nodes.min() < 0:
            raise ValueError("Invalid node id of {} or {}. Valid values are 0 to {}.".format(
                source_nodes.max(), source_nodes.min(), self.__nodes - 1))
        if self.__fg is None:
            self.__fg = numpy.zeros(self.__nodes, dtype=numpy.uint8)
        elif not self.__fg.flags.owndata:
            self.__fg = self.__fg.copy()   # (a view of the caller's mask, record_markers: never written through)
        if numpy.unique(source_nodes).size != source_nodes.size or self.__fg[source_nodes].any():
            for s in source_nodes:  # repeated ids accumulate in the reference; keep that via the explicit path
                self.set_tweight(int(s), self.MAX, 0)
            return
        self.__fg[source_nodes] = 1

    def set_sink_nodes(self, sink_nodes):
        sink_nodes = numpy.asarray(sink_nodes)
        if sink_nodes.size == 0:
            raise ValueError("max() arg is an empty sequence")
        if sink_nodes.max() >= self.__nodes or sink_nodes.min() < 0:
            raise ValueError("Invalid node id of {} or {}. Valid values are 0 to {}.".format(
                sink_nodes.max(), sink_nodes.min(), self.__nodes - 1))
        if self.__bg is None:
            self.__bg = numpy.zeros(self.__nodes, dtype=numpy.uint8)
        elif not self.__bg.flags.owndata:
            self.__bg = self.__bg.copy()
        if numpy.unique(sink_nodes).size != sink_nodes.size or self.__bg[sink_nodes].any():
            for s in sink_nodes:
                self.set_tweight(int(s), 0, self.MAX)
            return
        self.__bg[sink_nodes] = 1

    def set_nweight(self, node_from, node_to, weight_there, weight_back):
        if node_from >= self.__nodes or node_from < 0:
            raise ValueError("Invalid node id (node_from) of {}. Valid values are 0 to {}.".format(node_from, self.__nodes - 1))
        elif node_to >= self.__nodes or node_to < 0:
            raise ValueError("Invalid node id (node_to) of {}. Valid values are 0 to {}.".format(node_to, self.__nodes - 1))
        elif node_from == node_to:
            raise ValueError("The node_from ({}) can not be equal to the node_to ({}) (self-connections are forbidden in graph cuts).".format(node_from, node_to))
        elif weight_there <= 0 or weight_back <= 0:
            raise ValueError("Negative or zero weights are not allowed.")
        self.__edge_i.append(int(node_from))
        self.__edge_j.append(int(node_to))
        self.__edge_w.append(float(weight_there))
        self.__edge_r.append(float(weight_back))

    def set_nweights(self, nweights):
        for edge, weight in list(nweights.items()):
            self.set_nweight(edge[0], edge[1], weight[0], weight[1])

    def set_tweight(self, node, weight_source, weight_sink):
        if node >= self.__nodes or node < 0:
            raise ValueError("Invalid node id of {}. Valid values are 0 to {}.".format(node, self.__nodes - 1))
        if self.__tr is None:
            self.__tr = numpy.zeros(self.__nodes, dtype=numpy.float64)
        # Graph::add_tweights, graph.h:416-425, call by call
        cs, ck = float(weight_source), float(weight_sink)
        delta = self.__tr[node]
        if delta > 0:
            cs += delta
        else:
            ck -= delta
        self.__flow_const += cs if cs < ck else ck
        self.__tr[node] = cs - ck

    def set_tweights(self, tweights):
        for node, weight in list(tweights.items()):
            self.set_tweight(node, weight[0], weight[1])

    def set_tweights_all(self, tweights):
        for node, (twsource, twsink) in enumerate(tweights):
            self.set_tweight(node, twsource, twsink)

    def __edges_join_lattice_neighbours(self):
        """do all explicit edges join neighbours of the voxel lattice (of the neighbourhood in use)?  The tile solver keeps
        exactly those arcs; anything else (the reference accepts arbitrary node pairs, graph.py:382-440) goes to the
        sparse-graph solver."""
        shape = self.__shape if self.__lattice_shape is None else self.__lattice_shape
        if shape is None or not self.__edge_i:
            return True
        i = numpy.asarray(self.__edge_i, dtype=numpy.int64)
        j = numpy.asarray(self.__edge_j, dtype=numpy.int64)
        n = int(numpy.prod(shape))
        if i.max() >= n or j.max() >= n:
            return False
        ci = numpy.stack(numpy.unravel_index(i, shape))
        cj = numpy.stack(numpy.unravel_index(j, shape))
        d = numpy.abs(ci - cj)
        full = self.__connectivity not in (None, 2 * len(shape))
        ok = (d.max(axis=0) == 1) if full else (d.sum(axis=0) == 1)
        return bool(ok.all())

    def get_graph(self):
        """Builds the residual lattice in HBM (once) and returns the solver object."""
        if self.__graph is None and not self.__general and self.__edge_i and len(self.__shape or ()) <= 3 \
                and not self.__edges_join_lattice_neighbours():
            self.__general = True  # a plug-in term added an edge between voxels that are not neighbours
        if self.__graph is None and self.__general:
            self.__graph = self.__sparse_graph()
        if self.__graph is None and self.__lattice_shape is not None:
            self.__graph = EmbeddedLatticeGraph(self.__nodes, self.__lattice_shape, self.__boundary, self.__fg, self.__bg,
                                                device=self.__device, connectivity=self.__connectivity)
        if self.__graph is None:
            g = VoxelGraph(self.__shape, device=self.__device, connectivity=self.__connectivity)
            if self.__boundary is not None:
                g._set_boundary(*self.__boundary)
            if self.__regional is not None:
                g._set_regional(*self.__regional)
            if self.__tr is not None:
                g._set_tweights_merged(self.__tr, self.__flow_const)
            if self.__fg is not None or self.__bg is not None:
                g._set_markers(None if self.__fg is None else self.__fg, None if self.__bg is None else self.__bg)
            if self.__edge_i:
                g._add_edges(self.__edge_i, self.__edge_j, self.__edge_w, self.__edge_r)
            g._build()
            self.__graph = g
        return self.__graph

    def __sparse_graph(self):
        """the same calls in the same order (regional term, boundary term, explicit edges, markers last:
        generate.py:159-172, 322-338) on the sparse-graph solver"""
        if self.__connectivity not in (None, 2 * len(self.__shape)):
            raise NotImplementedError("medpy_amd: the full neighbourhood exists for 1-D..3-D voxel lattices only")
        g = SparseGraph(self.__nodes, self.__edges, device=self.__device)
        if self.__regional is not None:  # energy_voxel.py:61-65 in the map's dtype, then graph.py:551-552
            pm, alpha = self.__regional
            pm = numpy.asarray(pm)
            if pm.dtype not in (numpy.float32, numpy.float64):
                pm = pm.astype(numpy.float64)
            self.merge_tweights(numpy.arange(self.__nodes), (pm * alpha).ravel().astype(numpy.float64),
                                ((1 - pm) * alpha).ravel().astype(numpy.float64))
            self.__regional = None
        if self.__boundary is not None:
            g._add_lattice_edges(*self.__boundary)
        for term, lab, img, param in self.__label_terms:
            g._add_label_edges(term, lab, img, param)
        if self.__edge_i:
            g._add_edges(self.__edge_i, self.__edge_j, self.__edge_w, self.__edge_r)
        for marks, src, snk in ((self.__fg, float(self.MAX), 0.0), (self.__bg, 0.0, float(self.MAX))):
            if marks is not None:
                ids = numpy.flatnonzero(numpy.asarray(marks).ravel())
                self.merge_tweights(ids, numpy.full(ids.size, src), numpy.full(ids.size, snk))
        self.__fg = self.__bg = None
        if self.__tr is not None:
            g._set_tweights_merged(self.__tr, self.__flow_const)
        return g

    def get_node_count(self):
        return self.__nodes

    def get_nodes(self):
        return list(range(0, self.__nodes))

    def get_edge_count(self):
        return self.__edges
