import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_case(name):
    """-> (kind, inputs dict, weights dict, outputs dict) of a golden npz."""
    z = np.load(GOLDEN / f"{name}.npz", allow_pickle=False)
    inputs = {k[3:]: z[k] for k in z.files if k.startswith("in/")}
    w = {k[2:]: z[k] for k in z.files if k.startswith("w/")}
    outs = {k[4:]: z[k] for k in z.files if k.startswith("out/")}
    return str(z["meta/kind"]), inputs, w, outs


@pytest.fixture(scope="session")
def golden():
    return load_case


def assert_close(actual, expected, rel=1e-5, what=""):
    """|a-e| <= rel * max(|e|_inf, tiny) elementwise: 'within 1e-5 relative fp32' of the tensor's
    scale (BASELINE.json north_star), robust to elements that cancel to ~0."""
    a = np.asarray(actual, dtype=np.float64)
    e = np.asarray(expected, dtype=np.float64)
    assert a.shape == e.shape, f"{what}: shape {a.shape} != {e.shape}"
    if e.size == 0:
        return
    scale = max(float(np.max(np.abs(e))), 1e-30)
    err = float(np.max(np.abs(a - e)))
    assert np.isfinite(a).all(), f"{what}: non-finite values"
    assert err <= rel * scale, f"{what}: max abs err {err:.3e} > {rel:g} * scale {scale:.3e} (rel {err / scale:.3e})"
