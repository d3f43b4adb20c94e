"""Builds libmedpyhip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmedpyhip.so")
SOURCES = ["mgc_kernels.hip", "msg_sparse.hip"]
DEPS = ["mgc_kernels.hip", "msg_sparse.hip", "msg_node_ops.inl", "mgc_tile_ops.inl", "mgc_tile_ops26.inl", "mgc_wave_ops.inl", "mgc_wave_ops26.inl", "mgc_dt_ops.inl", "mgc_brick_ops.inl", "mgc_terms.h", "mgc_driver.inl", "mgc_common.h", os.path.join("..", "..", "include", "medpy_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-ldl"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    return all(os.path.getmtime(os.path.join(CSRC, d)) <= t for d in DEPS)


def build_library(force=False, verbose=False):
    if not force and up_to_date():
        return LIB
    cmd = [hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_library(force=True, verbose=True)
