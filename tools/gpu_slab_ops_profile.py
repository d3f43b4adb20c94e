"""Where does the time of a time-multiplexed two-slab solve go? (development aid)  Wall time per HipSlab method and per
solver op of the Python schedule (medpy_amd/slab.py:solve_slabs), every call synchronised by the library itself.

    python tools/gpu_slab_ops_profile.py [Z] [XY] [nslabs]
"""
import collections
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from medpy_amd import slab as slabmod  # noqa: E402
from medpy_amd.slab import HipSlab, LoopbackExchange, solve_slabs  # noqa: E402

Z = int(sys.argv[1]) if len(sys.argv) > 1 else 512
XY = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2
img, fg, bg = bench.block_volume(0, Z, Z // 512, XY // 512, 512)
acc, cnt = collections.Counter(), collections.Counter()


def timed(cls, name, key=None):
    f = getattr(cls, name)

    def g(self, *a, **k):
        t0 = time.perf_counter()
        r = f(self, *a, **k)
        kk = name if key is None else "%s[%s]" % (name, key(a))
        acc[kk] += time.perf_counter() - t0
        cnt[kk] += 1
        return r
    setattr(cls, name, g)


for m in ("halo_pack", "halo_unpack", "read_counts", "build", "finish_device"):
    timed(HipSlab, m, (lambda a: "kind %d" % a[1]) if m.startswith("halo") else None)
timed(HipSlab, "op", lambda a: str(a[0]))
slabs = [HipSlab(img.shape, r, N) for r in range(N)]
for s in slabs:
    sl = slice(s.plane0, s.plane1)
    s.set_boundary("difference_exponential", img[sl], 15.0, False)
    s.set_markers(fg[sl], bg[sl])
ex = LoopbackExchange(slabs)
for rep in range(2):
    acc.clear(); cnt.clear()
    t0 = time.perf_counter()
    for s in slabs:
        s.build()
    st = solve_slabs(slabs, ex)
    fl = sum(s.finish_device() for s in slabs)
    dt = time.perf_counter() - t0
print(json.dumps({"shape": list(img.shape), "slabs": N, "ms": round(dt * 1e3, 1), "flow": fl, **st}))
for k, v in acc.most_common():
    print("%-28s calls %6d  total %8.1f ms  avg %7.1f us" % (k, cnt[k], v * 1e3, v / cnt[k] * 1e6))
print("accounted: %.1f ms of %.1f" % (sum(acc.values()) * 1e3, dt * 1e3))
