"""The slab schedule inside the library (mgc_solve_slab) against the same schedule driven from Python (medpy_amd/slab.py), both
over the library's RCCL transport code with the in-process stand-in for librccl (tests/hostsim/mock_rccl.cpp): N ranks as
threads on ONE GPU (real RCCL refuses two ranks on one device).  Development aid; one JSON line per driver.

    python tools/gpu_slab_native.py [shape=512x512x512] [ranks=2] [conn=6]
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "hostsim"))
import sim  # noqa: E402

shape = sys.argv[1] if len(sys.argv) > 1 else "512x512x512"
ranks = sys.argv[2] if len(sys.argv) > 2 else "2"
conn = sys.argv[3] if len(sys.argv) > 3 else "6"
env = dict(os.environ, MEDPY_HIP_RCCL=sim.build_mock_rccl())
with tempfile.TemporaryDirectory() as tmp:
    for driver in ("slab.py", "native", "slab.py", "native"):
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "hostsim", "mock_rccl_worker.py"), ROOT, ranks, conn, "sphere", shape,
                              os.path.join(tmp, "labels.npy"), driver, "6" if conn == "26" else "8"], env=env, capture_output=True, text=True, timeout=900)
        line = res.stdout.strip().splitlines()[-1] if res.stdout.strip() else res.stderr[-500:]
        try:
            info = json.loads(line)
            st = info["stats"]
            print(json.dumps({"shape": shape, "ranks": int(ranks), "conn": int(conn), "driver": driver, "flow": info["flow"],
                              "solve_ms_per_rank": [s.get("solve_ms") for s in st],
                              **{k: st[0].get(k) for k in ("outer", "relabel_passes", "phases", "exchanges", "reductions", "readbacks", "converged") if k in st[0]}}), flush=True)
        except Exception:  # noqa: BLE001
            print(json.dumps({"driver": driver, "error": line[-400:]}), flush=True)
