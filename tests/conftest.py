import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_reference: needs /root/reference (only in the build container)")


def pytest_collection_modifyitems(config, items):
    from oracle.overlay import reference_present
    if reference_present():
        return
    skip = pytest.mark.skip(reason="/root/reference not present")
    for item in items:
        if "needs_reference" in item.keywords:
            item.add_marker(skip)


class Golden:
    """Lazy view on a tests/golden/*.npz written by oracle/gen_golden.py (reference outputs)."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name))

    def __getitem__(self, key):
        return self._z[key]

    def group(self, prefix):
        p = prefix + "/"
        return {k[len(p):]: self._z[k] for k in self._z.files if k.startswith(p) and "/" not in k[len(p):]}

    def has(self, key):
        return key in self._z.files


@pytest.fixture(scope="session")
def golden_kat():
    return Golden("reference_kat.npz")


@pytest.fixture(scope="session")
def golden_small():
    return Golden("reference_small.npz")


@pytest.fixture(scope="session")
def golden_synth():
    return Golden("reference_synthetic.npz")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    from oracle import bk
    bk.build()


# ---- the golden vectors a second time through the kernel forms a LARGE volume uses ------------------------------------
# k_discharge_w (the kernel the bench times) only takes over from 512 active tiles per colour phase on, and the activation
# only trusts the tiles' status words when there are more than 4096 candidates: a 64^3 fixture never gets there.  Every test
# of these modules therefore runs twice on the GPU: as shipped, and with the thresholds at zero (MEDPY_HIP_PARAMS is applied to
# every lattice handle at creation, medpy_amd/_lib.py:apply_env_params) plus the two-voxels-per-thread 26-neighbourhood
# discharge (wave_kernels bit 4).
_FORMS_MODULES = ("test_gpu_parity", "test_gpu_edge_cases", "test_gpu_slabs", "test_gpu_full_neighbourhood")
LARGE_VOLUME_FORMS = "wave_min_tiles=0,activate_exact_max=0,wave_kernels=25,exact_sink_tiles=2"
STORED_LABEL_FORMS = "wave_min_tiles=0,exact_sink_tiles=0,repeat_steps=2"  # the wave discharge without the exact labelling of tiles that hold a sink link;
# repeat_steps = 2: the instance with repeated in-plane steps exactly where the default uses the other one (graphs with walls) and vice versa
WAVE26_FORMS = "wave_kernels=41,prepush=0"  # 26-neighbourhood: the one-wave-per-tile discharge (k26_discharge_w), graph as built (no pre-push)


def pytest_generate_tests(metafunc):
    if metafunc.module.__name__.rsplit(".", 1)[-1] in _FORMS_MODULES and "kernel_forms" in metafunc.fixturenames:
        forms = ["as_shipped", "large_volume_forms"]
        if metafunc.module.__name__.rsplit(".", 1)[-1] == "test_gpu_parity":
            forms.append("stored_label_forms")
        if metafunc.module.__name__.rsplit(".", 1)[-1] == "test_gpu_full_neighbourhood":
            forms.append("wave26_forms")
        metafunc.parametrize("kernel_forms", forms, indirect=True)


@pytest.fixture(autouse=True)
def kernel_forms(request, monkeypatch):
    mode = getattr(request, "param", "as_shipped")
    if mode == "large_volume_forms":
        monkeypatch.setenv("MEDPY_HIP_PARAMS", LARGE_VOLUME_FORMS)
    elif mode == "stored_label_forms":
        monkeypatch.setenv("MEDPY_HIP_PARAMS", STORED_LABEL_FORMS)
    elif mode == "wave26_forms":
        monkeypatch.setenv("MEDPY_HIP_PARAMS", WAVE26_FORMS)
    return mode


@pytest.fixture(autouse=True, scope="session")
def _parity_log():
    """where oracle/cutcheck.py records every comparison that needed the tie relaxation (GPU box: under gpurun_out/)"""
    if "MEDPY_PARITY_LOG" not in os.environ and os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        os.environ["MEDPY_PARITY_LOG"] = os.path.join(ROOT, "gpurun_out", "parity_relaxations.jsonl")
    yield
