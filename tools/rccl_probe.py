import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from medpy_amd import synthetic
from medpy_amd.slab import HipSlab
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    print("torch imported", torch.__version__)
s = HipSlab((16, 16, 16), 0, 1)
uid = s.comm_unique_id()
print("uid ok", len(uid))
s.comm_init(uid)
print("comm init ok")
s.set_boundary("difference_linear", np.zeros((16, 16, 16), np.float32), None); s.build()
print(s.allreduce_counts())
