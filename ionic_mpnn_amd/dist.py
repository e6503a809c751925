"""Batch-sharded execution over the GPUs of one node: one process per GPU,
``torch.distributed`` with backend "nccl" (= RCCL over xGMI on ROCm); "gloo" for CPU rehearsal.

The reference has no distributed code at all; this is new.  Every (cation, anion[, T]) sample is
independent through the whole forward, so the data path needs NO collective: rank r takes the
contiguous rows [r*B/W, (r+1)*B/W) of every input (ionic_mpnn_amd.data.shard_bounds) and weights
are replicated.  Two optional epilogue collectives exist for callers that want them:

  * all_gather_fingerprints - the full (B, F) fingerprint / prediction matrix on every rank
    (per-sample rows, so all-gather is the right op);
  * all_reduce_loss_stats   - sum of squared error + sample count (2 scalars) for the MSE.
  * shard_loss_weight / all_reduce_flat_gradients_ - data-parallel training: loss weight of a shard and the
    single all-reduce of the flat gradient buffer (ionic_mpnn_amd.train).

Payloads are <= ~1 MB/rank (B=65536, D=32), i.e. latency-bound on xGMI's point-to-point links; one
fused buffer per call, RCCL picks its direct algorithms at this size.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .data import shard_bounds, shard_inputs  # noqa: F401  (re-exported)


def env_world():
    """(rank, local_rank, world_size) from the torchrun environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend=None, timeout_s=600):
    """Initialises the default process group from RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT.
    Returns (rank, local_rank, world_size).  No-op for world_size == 1."""
    import datetime
    rank, local_rank, world = env_world()
    if world == 1:
        return rank, local_rank, world
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # (HSA_ENABLE_IPC_MODE_LEGACY=0 - dmabuf IPC, which RCCL needs on this driver - only acts when it is in the
    #  environment before the process's FIRST GPU call: launchers export it, bench.py sets it before importing torch.
    #  Setting it here, after the caller has touched the device, would do nothing, so this function does not pretend to.)
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=timeout_s))
    return rank, local_rank, world


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def all_gather_fingerprints(local, group=None):
    """(B_local, F) per rank -> (sum B_local, F) on every rank, rank-major (= original row order
    for a contiguous shard).  Ranks may hold different row counts (B not divisible by W)."""
    if not is_distributed():
        return local
    world = dist.get_world_size(group)
    local = local.contiguous()
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    if len(set(counts)) == 1:
        out = torch.empty((world * counts[0], *local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    mx = max(counts)
    padded = torch.zeros((mx, *local.shape[1:]), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    bufs = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(bufs, padded, group=group)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def all_reduce_loss_stats(pred, target, group=None):
    """Global MSE pieces: returns (sum of squared error, count) reduced over all ranks as a
    float64 2-vector on pred's device (one collective)."""
    diff = (pred.reshape(-1).to(torch.float64) - target.reshape(-1).to(torch.float64))
    stats = torch.stack([torch.sum(diff * diff), torch.tensor(float(diff.numel()), dtype=torch.float64,
                                                              device=pred.device)])
    if is_distributed():
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats


def all_reduce_sum_(t, group=None):
    if is_distributed():
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def shard_loss_weight(n_local, device="cpu", group=None):
    """Data-parallel training: every rank holds ``n_local`` samples of the global
    mini-batch and computes the MEAN loss of its shard.  Scaling that loss by n_local / n_global before
    backward() makes the SUM of the ranks' gradients the gradient of the global mean - also for uneven
    shards and for penalties that every rank adds in full.  Returns (weight, n_global); one all-reduce."""
    cnt = torch.tensor([float(n_local)], dtype=torch.float64, device=device)
    if is_distributed():
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=group)
    n_global = float(cnt.item())
    return (float(n_local) / n_global if n_global > 0 else 0.0), int(n_global)


def all_reduce_flat_gradients_(flat, group=None):
    """One collective for every gradient of the model: the optimizer keeps them in one flat buffer."""
    if is_distributed():
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


class ShardedForward:
    """Runs ``fn(local_inputs) -> (B_local, F)`` on this rank's contiguous shard of a global batch."""

    def __init__(self, fn, world_size=None, rank=None):
        r, _, w = env_world()
        self.fn = fn
        self.world_size = w if world_size is None else world_size
        self.rank = r if rank is None else rank

    def local_inputs(self, global_inputs):
        return shard_inputs(global_inputs, self.world_size, self.rank)

    def __call__(self, global_inputs, gather=True):
        out = self.fn(self.local_inputs(global_inputs))
        return all_gather_fingerprints(out) if gather else out
