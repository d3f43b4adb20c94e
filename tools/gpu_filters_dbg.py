import sys, os, json, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MGC_DEBUG_SUSPECT"] = "1"
import numpy as np
from medpy_amd import synthetic, _lib
from medpy_amd.graphcut.graph import VoxelGraph
n = 256
s = synthetic.sphere((n, n, n))
g = VoxelGraph((n, n, n))
g._set_boundary("difference_exponential", s["image"], s["sigma"], False)
g._set_markers(s["fg"], s["bg"])
g.set_param("use_filters", 7)
g.set_param("max_outer", 4)
g._build()
try:
    g.maxflow()
except Exception as e:
    print("stopped:", str(e)[:80])
out = np.zeros(32, np.int32)
_lib.check(g._h, _lib.load().mgc_read_counts(g._h, _lib.ptr(out)))
print("counts", out.tolist())
