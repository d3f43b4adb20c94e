"""TEST ONLY (run as a subprocess by tests/test_gpu_slabs.py): N threads = N ranks on ONE GPU, each driving its own slab
handle through the library's own schedule and native transport (mgc_solve_slabs over mgc_comm_init / mgc_halo_exchange / ncclAllReduce / the
carry planes of the distance transforms by ncclSend / ncclRecv) with the
in-process mock of librccl (tests/hostsim/mock_rccl.cpp, selected by MEDPY_HIP_RCCL).  Prints one JSON line."""
import json
import os
import sys
import threading
import time

import numpy as np

root, nranks, conn, gen = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
shape = tuple(int(v) for v in sys.argv[5].split("x"))
ROUNDS = int(sys.argv[8]) if len(sys.argv) > 8 else 2  # colour rounds between two global relabels (the tests: 2, many relabels on small volumes)
sys.path.insert(0, root)
from medpy_amd import synthetic  # noqa: E402
from medpy_amd.slab import HipSlab, LoopbackExchange, solve_slabs, sync_boundary_table  # noqa: E402


class ThreadRcclExchange(object):
    """RcclExchange without a store: the unique id is handed over in-process"""

    native = True

    def __init__(self, slab, id_bytes):
        self.slabs = [slab]
        slab.comm_init(id_bytes)


s = getattr(synthetic, gen)(shape)
slabs = [HipSlab(shape, r, nranks, connectivity=conn) for r in range(nranks)]
uid = slabs[0].comm_unique_id()
out, errs = [None] * nranks, []
for sl in slabs:  # the term-by-table decision is one for the whole volume (integer-valued images): before any slab builds
    sl.set_boundary(s["term"], s["image"][sl.plane0:sl.plane1], s["sigma"], False)
sync_boundary_table(slabs, LoopbackExchange(slabs))


def run(r):
    try:
        sl = slabs[r]
        z = slice(sl.plane0, sl.plane1)
        sl.set_markers(s["fg"][z], s["bg"][z])
        sl.build()
        t0 = time.perf_counter()
        st = solve_slabs([sl], ThreadRcclExchange(sl, uid), rounds_per_relabel=ROUNDS)
        st["solve_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
        lab, part = sl.finish()
        out[r] = (lab, part, st)
    except Exception as e:  # noqa: BLE001
        errs.append("rank %d: %r" % (r, e))


threads = [threading.Thread(target=run, args=(r,), daemon=True) for r in range(nranks)]
for t in threads:
    t.start()
for t in threads:
    t.join(240)
if errs or any(t.is_alive() for t in threads):
    print(json.dumps({"error": errs or ["a rank did not finish (protocol deadlock)"]}))
    sys.stdout.flush()
    os._exit(1)
labels = np.concatenate([o[0] for o in out], axis=0)
np.save(sys.argv[6], labels)
print(json.dumps({"flow": float(sum(o[1] for o in out)), "stats": [o[2] for o in out]}))
sys.stdout.flush()
os._exit(0)
