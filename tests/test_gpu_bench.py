"""bench.py as the driver launches it: the one-JSON-line contract on one GPU, and the N > 1 command path
(`python -m torch.distributed.run ... bench.py --gpus N`) rehearsed with two ranks that share this box's GPU over gloo -
incl. `--config4` (BASELINE.json configs[3] as worded: the all-gather of the pooled fingerprints and the all-reduce of
the loss statistics inside the timed loop).  Child processes, so that a rank never re-executes a GPU-initialised parent."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, timeout=420):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, f"{' '.join(cmd)}\n--- stdout\n{p.stdout[-2000:]}\n--- stderr\n{p.stderr[-4000:]}"
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, f"exactly one JSON line expected, got {len(lines)}"
    return json.loads(lines[0])


def test_bench_line_contract_one_gpu():
    out = _run([sys.executable, "bench.py", "--steps", "5", "--warmup", "2", "--ramp-ms", "20", "--batch", "512",
                "--no-cpu-baseline", "--no-other-configs"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 5 and out["value"] > 0 and out["vs_baseline"] is None
    assert out["dtype"] in ("f32", "f32 (bf16x9 emulation)")
    assert set(out["config"]["modes_timed"]) == {"f32t", "f32x3"}          # both exact-f32 forms, same K steps
    best = max(out["config"]["modes_timed"], key=lambda k: out["config"]["modes_timed"][k]["graph_pairs_per_s"])
    assert out["config"]["mode"] == best                                     # f32x3 carries `value` only where it wins
    assert (out["dtype"] == "f32") == (best == "f32t")
    assert "prepared_weights" in out["config"] and out["config"]["prepared_weights"]["us_per_weight_version"] > 0
    r = out["roofline"]
    assert r["bound"] == "mfma" and 0 < r["frac"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9


@pytest.mark.parametrize("extra", [[], ["--config4"]])
def test_two_rank_gloo_rehearsal_of_the_driver_command(extra):
    """Two ranks on ONE GPU (gloo): the exact N > 1 command path of the driver; not a scaling number."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1",
           "--ramp-ms", "0", "--batch", "256", "--no-cpu-baseline", "--no-other-configs", *extra]
    out = _run(cmd)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 512                              # whole-job aggregate: 256 pairs per rank
    assert "rehearsal" in out["config"]
    if extra:
        assert out["config"]["gathered_fingerprint_rows"] == 512             # every rank holds all fingerprints
        assert out["config"]["collective_us"] is not None
        assert out["config"]["without_collectives"]["graph_pairs_per_s"] > 0
        assert "configs[3]" in out["config"]["workload"]


def test_config5_training_bench_two_rank_rehearsal():
    """`bench.py --config5 --gpus 2` (BASELINE.json configs[4] as worded: the data-parallel training step with its gradient
    all-reduce inside the timed loop), two ranks on one GPU over gloo, small batch."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--batch", "64", "--config5"]
    out = _run(cmd)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert out["config"]["global_batch"] == 128 and "configs[4]" in out["config"]["workload"]
    assert "rehearsal" in out["config"] and out["roofline"]["bound"] == "mfma"
    assert np.isfinite(out["config"]["loss_last"])
