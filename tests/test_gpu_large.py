"""-m gpu: the sizes the metric is quoted on, voxel for voxel.  tests/golden/reference_large.json holds the SHA-256 of the
packed label volume, the flow and the foreground count the UNMODIFIED reference solver (oracle/_ref, compiled from
/root/reference) produces for each case (oracle/gen_golden.py large); the oracle itself needs minutes and tens of GB
there, so the GPU box only compares hashes.  BASELINE.json configs: [1] 256^3 6-conn, headline 512^3 6-conn, [2]'s shape of
problem (26-conn + regional term) at 256^3."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_large.json")
_CASES = json.load(open(_PATH)) if os.path.exists(_PATH) else {}


@pytest.mark.parametrize("name", sorted(_CASES))
def test_large_volume_labels_hash(name):
    from medpy_amd import graphcut, synthetic
    c = _CASES[name]
    shape = tuple(c["shape"])
    s = getattr(synthetic, c["gen"])(shape)
    kw = dict(boundary_term=graphcut.energy_voxel.boundary_difference_exponential, boundary_term_args=(s["image"], s["sigma"], False))
    if c["regional"]:
        r = synthetic.regional(shape)
        kw.update(regional_term=graphcut.energy_voxel.regional_probability_map, regional_term_args=(r["prob"], r["alpha"]))
    if c["connectivity"] != 6:
        kw["connectivity"] = c["connectivity"]
    g = graphcut.graph_from_voxels(s["fg"], s["bg"], **kw)
    flow = g.maxflow()
    labels = g.labels()
    assert int(labels.sum()) == c["foreground_voxels"]
    assert hashlib.sha256(np.packbits(labels.astype(np.uint8).ravel()).tobytes()).hexdigest() == c["sha256_packed_labels"]
    assert flow == pytest.approx(c["flow"], rel=1e-9)
    print(name, "flow", flow, "stats", g.stats())
