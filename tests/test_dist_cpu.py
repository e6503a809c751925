"""CPU, world_size 2, gloo: the batch-shard path - contiguous shard bounds, rank-major
all-gather of per-sample rows (uneven shards included) and the 2-scalar loss all-reduce.  The compute
itself is GPU-only; here each rank's "forward" is a deterministic per-row function so that the test
checks exactly what the distributed layer adds: partitioning and reassembly."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_rows, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from ionic_mpnn_amd import dist as idist
    from ionic_mpnn_amd import synthetic
    r, lr, w = idist.init_distributed(backend="gloo")
    assert (r, w) == (rank, world) and idist.is_distributed()
    glob = synthetic.make_batch(n_rows, max_atoms=6, max_edges=10, atom_vocab_size=9, bond_vocab_size=4, min_atoms=2,
                                seed=3)

    def fake_forward(local):  # per-sample, independent of the rest of the batch - like the real forward
        a = torch.from_numpy(local["cat_atom"]).float()
        c = torch.from_numpy(local["an_connectivity"]).float()
        return torch.stack([a.sum(1), (a * a).sum(1), c.sum((1, 2)), torch.from_numpy(local["temperature"])[:, 0]], 1)

    sf = idist.ShardedForward(fake_forward)
    lo, hi = idist.shard_bounds(n_rows, world, rank)
    local = sf.local_inputs(glob)
    assert len(local["cat_atom"]) == hi - lo and (local["cat_bond"] == glob["cat_bond"][lo:hi]).all()
    full = sf(glob)                                   # all-gather, rank-major
    ref = fake_forward(glob)
    assert full.shape == ref.shape and torch.equal(full, ref)
    # loss statistics: one all-reduce of (sum sq err, count)
    target = torch.arange(n_rows, dtype=torch.float32)
    stats = idist.all_reduce_loss_stats(sf(glob, gather=False)[:, 0], target[lo:hi])
    sse = float(((ref[:, 0].double() - target.double()) ** 2).sum())
    assert abs(float(stats[0]) - sse) <= 1e-9 * max(1.0, sse) and int(stats[1]) == n_rows
    t = idist.all_reduce_sum_(torch.tensor([float(rank + 1)]))
    assert float(t) == world * (world + 1) / 2
    Path(out_dir, f"ok_{rank}").write_text("ok")
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_rows", [8, 7])            # even and uneven shards
def test_world2_gloo_shard_gather_reduce(tmp_path, n_rows):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_rows, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    for r, p in enumerate(procs):
        if p.is_alive():
            p.kill()
            pytest.fail(f"rank {r} hung")
        assert p.exitcode == 0, f"rank {r} exit code {p.exitcode}"
        assert (tmp_path / f"ok_{r}").exists()


def test_single_process_is_a_noop():
    sys.path.insert(0, str(ROOT))
    from ionic_mpnn_amd import dist as idist
    assert idist.env_world()[2] >= 1
    x = torch.arange(6.0).reshape(3, 2)
    if not idist.is_distributed():
        assert idist.all_gather_fingerprints(x) is x
        s = idist.all_reduce_loss_stats(x[:, 0], torch.zeros(3))
        assert float(s[0]) == float((x[:, 0] ** 2).sum()) and int(s[1]) == 3


def _grad_worker(rank, world, port, n_rows, out_dir):
    sys.path.insert(0, str(ROOT))
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    from ionic_mpnn_amd import dist as idist
    idist.init_distributed(backend="gloo")
    g = torch.Generator().manual_seed(0)
    X, y = torch.randn(n_rows, 5, generator=g, dtype=torch.float64), torch.randn(n_rows, generator=g, dtype=torch.float64)
    w0 = torch.randn(5, generator=g, dtype=torch.float64)
    lam = 0.3

    def loss_of(w, Xs, ys):  # mean loss of a shard + a penalty every rank adds in full (like the l2 term)
        return torch.mean((Xs @ w - ys) ** 2) + lam * (w * w).sum()

    wf = w0.clone().requires_grad_(True)
    loss_of(wf, X, y).backward()                      # full-batch gradient: the target
    lo, hi = idist.shard_bounds(n_rows, world, rank)
    wl = w0.clone().requires_grad_(True)
    flat = torch.zeros(5, dtype=torch.float64)
    wl.grad = flat.view_as(wl)                        # gradients accumulate into the flat buffer, as in train.Adam
    weight, n_glob = idist.shard_loss_weight(hi - lo)
    assert n_glob == n_rows
    if hi > lo:
        (loss_of(wl, X[lo:hi], y[lo:hi]) * weight).backward()
    idist.all_reduce_flat_gradients_(flat)
    assert torch.allclose(flat, wf.grad, rtol=1e-12, atol=1e-12), (flat, wf.grad)
    Path(out_dir, f"gok_{rank}").write_text("ok")
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_rows", [8, 7, 1])         # even, uneven, and a rank with an empty shard
def test_world2_gloo_gradient_average_matches_full_batch(tmp_path, n_rows):
    world, port = 2, _free_port()
    mp.spawn(_grad_worker, args=(world, port, n_rows, str(tmp_path)), nprocs=world, join=True)
    assert all((tmp_path / f"gok_{r}").exists() for r in range(world))


def test_early_stopping_and_history_host_logic():
    from ionic_mpnn_amd import train

    class Fake:
        def __init__(self):
            self.w = 0
        def state_dict(self):
            return {"w": self.w}
        def load_weights(self, d):
            self.w = d["w"]

    m, es, hist = Fake(), train.EarlyStopping(monitor="val_loss", patience=2, restore_best_weights=True), train.History()
    m.stop_training = False
    es.set_model(m)
    es.on_train_begin()
    stopped = None
    for epoch, vl in enumerate([3.0, 2.0, 2.5, 2.2, 1.9, 5.0]):
        m.w = epoch
        hist._log(epoch, {"loss": vl + 1, "val_loss": vl})
        es.on_epoch_end(epoch, {"val_loss": vl})
        if m.stop_training:
            stopped = epoch
            break
    es.on_train_end()
    assert stopped == 3 and es.best == 2.0 and m.w == 1          # two epochs without improvement after epoch 1
    assert hist.history["val_loss"] == [3.0, 2.0, 2.5, 2.2] and hist.epoch == [0, 1, 2, 3]

    class Selective(train.Callback):                              # the reference's own callback ports as is
        def __init__(self):
            super().__init__()
            self.seen = []
        def on_epoch_end(self, epoch, logs=None):
            self.seen.append((epoch + 1, logs.get("loss", 0)))
    sv = Selective(); sv.set_model(m); sv.on_epoch_end(0, {"loss": 1.5})
    assert sv.seen == [(1, 1.5)] and sv.model is m
    a = train.Adam(1e-3, clipnorm=1.0)
    assert a.get_config()["clipnorm"] == 1.0 and a.get_config()["epsilon"] == 1e-7
