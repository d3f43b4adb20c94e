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
