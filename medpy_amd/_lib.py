"""ctypes binding of libmedpyhip.so (C ABI: include/medpy_hip.h).

There is no CPU fallback: if the HIP library is missing or no MI355X is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MEDPY_HIP_LIB") or os.path.join(_HERE, "libmedpyhip.so")  # override: development builds

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_OOM, ERR_STATE, ERR_UNSUPPORTED, ERR_NOT_CONVERGED = range(8)

TERM_IDS = {
    "none": 0,
    "difference_linear": 1, "difference_exponential": 2, "difference_division": 3, "difference_power": 4,
    "maximum_linear": 5, "maximum_exponential": 6, "maximum_division": 7, "maximum_power": 8,
}

DTYPE_IDS = {
    np.dtype(np.uint8): 0, np.dtype(np.int8): 1, np.dtype(np.uint16): 2, np.dtype(np.int16): 3,
    np.dtype(np.uint32): 4, np.dtype(np.int32): 5, np.dtype(np.uint64): 6, np.dtype(np.int64): 7,
    np.dtype(np.float32): 8, np.dtype(np.float64): 9,
}


class Stats(C.Structure):
    _fields_ = [
        ("build_ms", C.c_double), ("solve_ms", C.c_double), ("discharge_ms", C.c_double), ("relabel_ms", C.c_double),
        ("discharge_launches", C.c_int64), ("relabel_launches", C.c_int64), ("discharge_tiles", C.c_int64),
        ("relabel_tiles", C.c_int64), ("global_relabels", C.c_int64), ("phases", C.c_int64), ("ntiles", C.c_int64),
        ("nvox", C.c_int64), ("device_bytes", C.c_int64), ("reserved", C.c_int64 * 3),
        ("discharge_wave_ms", C.c_double), ("discharge_wave_launches", C.c_int64), ("discharge_wave_tiles", C.c_int64),
        ("timing_stride", C.c_int64),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}
        d["readbacks"] = self.reserved[0]
        d["radial_cycles"] = self.reserved[1]  # cycles of colour phases that ran on radial labels (mgc_driver.inl)
        d["wall_tiles"] = self.reserved[2]  # tiles a surface of weak arcs passes through, as built (MGC_WALL_*, mgc_common.h)
        return d


class Validation(C.Structure):
    """mgc_validation (include/medpy_hip.h): invariants of a maximum preflow, counted on the device"""
    _fields_ = [("voxels", C.c_int64), ("negative_values", C.c_int64), ("active_excess", C.c_int64),
                ("residual_arcs_across", C.c_int64), ("sink_links_across", C.c_int64), ("pair_violations", C.c_int64),
                ("node_violations", C.c_int64), ("pending_outbox", C.c_int64), ("reserved", C.c_int64 * 4),
                ("max_pair_error", C.c_double), ("max_node_error", C.c_double), ("flow_into_sink", C.c_double),
                ("cut_capacity", C.c_double), ("flow_constant", C.c_double), ("sink_capacity_used", C.c_double),
                ("reserved_d", C.c_double * 2)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("reserved")}


VIOLATION_KEYS = ("negative_values", "active_excess", "residual_arcs_across", "sink_links_across", "pair_violations",
                  "node_violations", "pending_outbox")


def assert_valid(v, rel=1e-9):
    """v: Validation.as_dict() (summed over the ranks for a volume cut into slabs).  Raises AssertionError naming the
    violated invariant; returns the relative difference between the flow into the sink and the capacity of the cut."""
    bad = {k: v[k] for k in VIOLATION_KEYS if v[k]}
    assert not bad, "max-flow invariants violated: %r" % (bad,)
    # The flow that reached the sink is a sum of differences (built sink link - residual): every push into a sink link of
    # 65535 rounds at 7e-12, so the sum is only known to ~1e-13 of the sink capacity in use (hundreds of pushes per link).
    # The capacity of the cut is exact; the zero counts above already imply flow == cut (every arc across the cut saturated,
    # conservation on the sink side).  This comparison is the independent, coarser cross-check.
    scale = max(abs(v["flow_into_sink"]), abs(v["cut_capacity"]), 1e-300)
    diff = abs(v["flow_into_sink"] - v["cut_capacity"])
    assert diff <= rel * scale + 1e-13 * v["sink_capacity_used"], "flow into the sink %r != capacity of the cut %r" % (
        v["flow_into_sink"], v["cut_capacity"])
    return diff / scale


class SlabStats(C.Structure):
    """mgc_slab_stats (include/medpy_hip.h): what mgc_solve_slab did"""
    _fields_ = [("outer", C.c_int64), ("relabel_passes", C.c_int64), ("phases", C.c_int64), ("exchanges", C.c_int64),
                ("reductions", C.c_int64), ("converged", C.c_int64), ("discharge_tiles", C.c_int64), ("relabel_tiles", C.c_int64),
                ("deferred_drains", C.c_int64), ("reserved", C.c_int64 * 7)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_ if k != "reserved"}


# mgc_transport (include/medpy_hip.h): the callbacks of a host transport for mgc_solve_slabs
XCHG_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.c_int, C.c_int)
SEND_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64)
RECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64)


class Transport(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("exchange", XCHG_FN), ("allreduce", ALLREDUCE_FN), ("send", SEND_FN), ("recv", RECV_FN)]


class SparseStats(C.Structure):
    _fields_ = [("build_ms", C.c_double), ("solve_ms", C.c_double), ("rounds", C.c_int64), ("global_relabels", C.c_int64),
                ("relabel_passes", C.c_int64), ("nodes", C.c_int64), ("arcs", C.c_int64), ("edges_added", C.c_int64),
                ("reserved", C.c_int64 * 4)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if k != "reserved"}


LABEL_TERM_IDS = {"stawiaski": 1, "stawiaski_directed": 2, "difference_of_means": 3}

# every symbol include/medpy_hip.h declares: (restype, argtypes)
_VP, _I64, _DBL, _INT = C.c_void_p, C.c_int64, C.c_double, C.c_int
SIGNATURES = {
    "mgc_device_count": (_INT, [C.POINTER(_INT)]),
    "mgc_device_memory": (_INT, [_INT, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mgc_create": (_INT, [_INT, C.POINTER(_I64), _INT, _INT, C.POINTER(_VP)]),
    "mgc_destroy": (_INT, [_VP]),
    "mgc_last_error": (C.c_char_p, [_VP]),
    "mgc_pool_trim": (_INT, [_INT]),
    "mgc_pool_info": (_INT, [_INT, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mgc_set_boundary": (_INT, [_VP, _INT, _VP, _INT, _DBL, C.POINTER(_DBL)]),
    "mgc_set_boundary_lut": (_INT, [_VP, _VP, _I64]),
    "mgc_set_regional_probability": (_INT, [_VP, _VP, _INT, _DBL]),
    "mgc_set_markers": (_INT, [_VP, _VP, _VP]),
    "mgc_add_edges": (_INT, [_VP, _I64, _VP, _VP, _VP, _VP]),
    "mgc_set_tweights_merged": (_INT, [_VP, _VP, _DBL]),
    "mgc_build": (_INT, [_VP]),
    "mgc_get_nweights": (_INT, [_VP, _INT, _VP]),
    "mgc_get_tweights": (_INT, [_VP, _VP]),
    "mgc_get_nweights_offset": (_INT, [_VP, C.POINTER(_INT), _VP]),
    "mgc_get_edge": (_INT, [_VP, _I64, _I64, C.POINTER(_DBL)]),
    "mgc_maxflow": (_INT, [_VP, C.POINTER(_DBL)]),
    "mgc_labels": (_INT, [_VP, _VP]),
    "mgc_what_segment": (_INT, [_VP, _I64, C.POINTER(_INT)]),
    "mgc_get_node_num": (_INT, [_VP, C.POINTER(_I64)]),
    "mgc_set_param": (_INT, [_VP, C.c_char_p, _I64]),
    "mgc_validate": (_INT, [_VP, _VP]),
    "mgc_get_image_range": (_INT, [_VP, _VP]),
    "mgc_set_image_range": (_INT, [_VP, _VP]),
    "mgc_get_stats": (_INT, [_VP, C.POINTER(Stats)]),
    "mgc_get_profile": (_INT, [_VP, _VP]),
    # Z-slab decomposition (multi-GPU)
    "mgc_create_slab": (_INT, [_INT, C.POINTER(_I64), _INT, _INT, _INT, _INT, C.POINTER(_VP)]),
    "mgc_slab_info": (_INT, [_VP, C.POINTER(_I64)]),
    "mgc_solver_op": (_INT, [_VP, _INT, _I64, _I64, _I64, _I64]),
    "mgc_read_counts": (_INT, [_VP, _VP]),
    "mgc_halo_bytes": (_INT, [_VP, _INT, C.POINTER(_I64)]),
    "mgc_halo_pack": (_INT, [_VP, _INT, _INT, _VP, _INT]),
    "mgc_halo_unpack": (_INT, [_VP, _INT, _INT, _VP, _INT, C.c_uint32, _INT]),
    "mgc_finish": (_INT, [_VP, C.POINTER(_DBL)]),
    "mgc_comm_unique_id": (_INT, [_VP]),
    "mgc_comm_init": (_INT, [_VP, _VP]),
    "mgc_halo_exchange": (_INT, [_VP, _INT, C.c_uint32, _INT]),
    "mgc_allreduce_counts": (_INT, [_VP, _VP]),
    "mgc_solve_slab": (_INT, [_VP, C.POINTER(SlabStats)]),
    "mgc_solve_slabs": (_INT, [C.POINTER(C.c_void_p), _INT, C.POINTER(Transport), C.POINTER(SlabStats)]),
    # sparse graphs (region graph cut, n-D voxel graphs, edge-by-edge plug-ins)
    "msg_create": (_INT, [_I64, _INT, C.POINTER(_VP)]),
    "msg_destroy": (_INT, [_VP]),
    "msg_last_error": (C.c_char_p, [_VP]),
    "msg_set_param": (_INT, [_VP, C.c_char_p, _I64]),
    "msg_add_edges": (_INT, [_VP, _I64, _VP, _VP, _VP, _VP]),
    "msg_add_lattice_edges": (_INT, [_VP, _INT, _INT, C.POINTER(_I64), _VP, _INT, _DBL, C.POINTER(_DBL)]),
    "msg_add_label_edges": (_INT, [_VP, _INT, _INT, C.POINTER(_I64), _VP, _VP, _INT, _DBL]),
    "msg_region_sums": (_INT, [_INT, _I64, _VP, _VP, _INT, _INT, _I64, _VP, _VP]),
    "msg_set_tweights_merged": (_INT, [_VP, _VP, _DBL]),
    "msg_maxflow": (_INT, [_VP, C.POINTER(_DBL)]),
    "msg_labels": (_INT, [_VP, _VP]),
    "msg_what_segment": (_INT, [_VP, _I64, C.POINTER(_INT)]),
    "msg_get_edge": (_INT, [_VP, _I64, _I64, C.POINTER(_DBL)]),
    "msg_get_counts": (_INT, [_VP, C.POINTER(_I64), C.POINTER(_I64), C.POINTER(_I64)]),
    "msg_get_arcs": (_INT, [_VP, _VP, _VP, _VP]),
    "msg_get_stats": (_INT, [_VP, C.POINTER(SparseStats)]),
}

_lib = None


class MedpyHipError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "libmedpyhip error %d: %s" % (code, message))
        self.code = code


def load():
    """Load the HIP library; raises ImportError with the build recipe when it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "medpy_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # RTLD_DEEPBIND: bind to the HIP runtime the library was linked against (system ROCm) even when the host
    # application (e.g. a PyTorch wheel, which bundles its own libamdhip64) already exported HIP symbols globally.
    lib = C.CDLL(LIB_PATH, mode=os.RTLD_NOW | os.RTLD_LOCAL | os.RTLD_DEEPBIND)
    for name, (res, args) in SIGNATURES.items():
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    _lib = lib
    return lib


def device_count():
    n = C.c_int(0)
    load().mgc_device_count(C.byref(n))
    return n.value


def device_free_bytes(device=0):
    """free HBM of a device in bytes, or None when it cannot be asked"""
    f, t, idle = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    if load().mgc_device_memory(int(device), C.byref(f), C.byref(t)) != OK:
        return None
    load().mgc_pool_info(int(device), C.byref(idle), None, None)  # what the library's own pool holds gives way to any allocation that needs it
    return f.value + idle.value


def pool_trim(device=0):
    """hand the device memory the library keeps for the next handle (mgc_pool_*) back to the driver"""
    return load().mgc_pool_trim(int(device))


def pool_info(device=0):
    idle, hits, misses = C.c_int64(0), C.c_int64(0), C.c_int64(0)
    load().mgc_pool_info(int(device), C.byref(idle), C.byref(hits), C.byref(misses))
    return {"idle_bytes": idle.value, "hits": hits.value, "misses": misses.value}


def check(handle, rc):
    if rc != OK:
        msg = load().mgc_last_error(handle)
        raise MedpyHipError(rc, (msg or b"").decode("utf-8", "replace"))


def apply_env_params(handle):
    """MEDPY_HIP_PARAMS="name=value,name=value": schedule / kernel-form knobs (mgc_set_param) applied to every lattice handle at
    creation.  How the test suite sends all its golden vectors through the kernel forms a large volume uses (the wave
    kernels only take over from 512 active tiles per phase on) without touching the tests themselves."""
    spec = os.environ.get("MEDPY_HIP_PARAMS", "")
    for kv in filter(None, (p.strip() for p in spec.split(","))):
        name, _, value = kv.partition("=")
        check(handle, load().mgc_set_param(handle, name.strip().encode(), int(value)))


def check_sparse(handle, rc):
    if rc != OK:
        msg = load().msg_last_error(handle)
        raise MedpyHipError(rc, (msg or b"").decode("utf-8", "replace"))


def ptr(a):
    return C.c_void_p(a.ctypes.data)
