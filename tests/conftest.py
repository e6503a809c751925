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


def assert_close(actual, expected, rel=1e-5, what="", floor=0.3):
    """'within 1e-5 relative fp32' (BASELINE.json north_star), checked two ways:
      * per tensor:   max|a-e| <= rel * max|e|;
      * per element:  |a_i - e_i| <= rel * max(|e_i|, floor * max|e|) - a true elementwise relative bound for every
        element that is not small against the tensor, and an absolute bound `floor` x tighter than the per-tensor
        one for the elements that cancel towards 0 (an f32 sum's error scales with its partial sums, not with its
        result, so no f32 implementation - the reference's included - meets a pure elementwise bound there)."""
    a = np.asarray(actual, dtype=np.float64)
    e = np.asarray(expected, dtype=np.float64)
    assert a.shape == e.shape, f"{what}: shape {a.shape} != {e.shape}"
    if e.size == 0:
        return
    scale = max(float(np.max(np.abs(e))), 1e-30)
    err = float(np.max(np.abs(a - e)))
    assert np.isfinite(a).all(), f"{what}: non-finite values"
    assert err <= rel * scale, f"{what}: max abs err {err:.3e} > {rel:g} * scale {scale:.3e} (rel {err / scale:.3e})"
    bound = rel * np.maximum(np.abs(e), floor * scale)
    worst = float(np.max(np.abs(a - e) / bound))
    assert worst <= 1.0, f"{what}: elementwise error {worst:.2f} x its bound (rel {rel:g}, floor {floor:g} of scale {scale:.3e})"
