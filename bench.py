#!/usr/bin/env python3
"""bench.py - graph-pairs/sec of the message-passing forward hot path on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (embedding gather, S x (BondMatrixMessage,
Reduce, GatedUpdate), GlobalSumPool, for the cation AND the anion branch) over one resident batch
of synthetic padded graph pairs: BASELINE.json configs[1] (N=40, E=80, D=32, K=8, S=3, batch 4096
per GPU; weak scaling: every rank owns its own 4096 pairs, no data-path collective).
`--config4`: BASELINE.json configs[3] as worded - 8192 pairs per rank, and every timed step ends with the all-gather
of the pooled fingerprints and the all-reduce of the loss statistics (RCCL); the no-collective figure stands beside it.
`--config5`: BASELINE.json configs[4] as worded - the full training step (fwd + bwd + Adam) at atom_dim 128, 6 steps,
data-parallel over the ranks, the gradient all-reduce inside every timed step (a line of its own: another metric).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

# the host driver supports dmabuf IPC only (RCCL / CUDA-tensor sharing across processes): must be in the environment
# BEFORE the first GPU call of the process, so it is set before torch is imported
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3    # MI355X_MICROARCH.md: dense f32 MFMA peak (16x16x4 / 4x4x1 f32: 256 flop/clk/CU)
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA peak
PEAK_HBM_GBS = 8000.0           # HBM3E spec


def algorithmic_flops_per_pair(N, E, D, S):
    """SURVEY.md 8(d), per-bond-type schedule: 2 ions * S * (2 D^2 E + 12 D^2 N)."""
    return 2 * S * (2 * D * D * E + 12 * D * D * N)


def algorithmic_bytes_per_pair(N, E, D):
    """SURVEY.md 8(d), fused forward: 2*(4N + 12E + 4D) (+8 for T and the output scalar)."""
    return 2 * (4 * N + 12 * E + 4 * D) + 8


def executed_counts(inp):
    """(kept rows, valid edges) over both ions: what the kernels multiply (padding atoms / edge slots are skipped
    exactly): update 12 D^2 per kept row, message 2 D^2 per valid edge and step."""
    rows = edges = 0
    for pfx in ("cat", "an"):
        ids, conn = inp[f"{pfx}_atom"], inp[f"{pfx}_connectivity"]
        ok = (conn[:, :, 0] > 0) & (conn[:, :, 1] > 0)
        last_id = np.where(ids > 0, np.arange(ids.shape[1])[None, :] + 1, 0).max(axis=1)
        last_e = np.where(ok, conn.max(axis=2) + 1, 0).max(axis=1)
        rows += int(np.maximum(last_id, last_e).sum())
        edges += int(ok.sum())
    return rows, edges


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    cap = os.environ.get("IMPNN_BENCH_CORES")  # optional override; default: every core this process may use
    return max(1, min(n, int(cap))) if cap else max(1, n)


def pmc_field(name, kernel="encoder_typed"):
    """A figure for the encoder kernel from the committed rocprofv3 --pmc passes (profiles/pmc_*.json,
    latest round wins), or None.  hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: the gfx950
    FETCH_SIZE correction of MI355X_MICROARCH.md's HBM section.  (PMC passes are separate rocprofv3 runs of this
    same command, tools/pmc_profile.sh; the figure is per launch of the named kernel.)"""
    best, src = None, None
    key = kernel if kernel in ("encoder_typed", "encoder_typed_x3") else ("encoder_typed" if "typed" in kernel else "encoder_fused")
    for f in sorted((ROOT / "profiles").glob("pmc_*.json")):
        try:
            v = json.loads(f.read_text()).get(key, {}).get(name)
        except (OSError, ValueError):
            v = None
        if v is not None:
            best, src = v, f.name
    return best, src


def _time_cpu(fn, warmup, iters):
    """median seconds of fn() over `iters` timed runs after `warmup` untimed ones"""
    for _ in range(warmup):
        fn()
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return float(np.median(times)), iters, warmup


def cpu_baseline(inputs, w, gpu_pooled=None):
    """Reference-schedule torch-CPU forward (oracle/torch_ref.py) of the same workload on the host cores this
    process may use: batch 32 (the reference's own batch, train_viscosity.py:332) and batch 4096
    (the GPU workload), each with 5 warm-up + 20 timed iterations, median (about 45 s of CPU work on the GPU box's 16
    cores: the (B,E,D,D) tensor the reference materialises is 1.3 GB per layer call at batch 4096).
    Reported beside the GPU number; never `value`.  A 256-pair sample of the GPU result of the timed configuration is
    also checked against the fp64 run of the port (BASELINE.json: "fp32 max-abs-err vs ref")."""
    from oracle import torch_ref as TR
    cores = usable_cores()
    torch.set_num_threads(cores)
    acc = None
    if gpu_pooled is not None:
        sample = 256
        sub = {k: v[:sample] for k, v in inputs.items()}
        ref = [t.numpy() for t in TR.pooled_pair(w, sub, torch.float64)]
        got = [t[:sample].double().cpu().numpy() for t in gpu_pooled]
        abs_err = max(float(np.abs(g - r).max()) for g, r in zip(got, ref))
        rel_err = max(float(np.abs(g - r).max() / np.abs(r).max()) for g, r in zip(got, ref))
        acc = {"max_abs_err": abs_err, "max_rel_err": rel_err,
               "of": f"GlobalSumPool outputs of the first {sample} pairs vs the fp64 CPU port (tolerance 1e-5 relative)"}
    B = int(next(iter(inputs.values())).shape[0])
    b32 = {k: v[:32] for k, v in inputs.items()}
    med32, n32, w32 = _time_cpu(lambda: TR.pooled_pair(w, b32), 5, 20)
    medB, nB, wB = _time_cpu(lambda: TR.pooled_pair(w, inputs), 5, 20)
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in Path("/proc/cpuinfo").read_text().splitlines()
                         if l.startswith("model name"))
    except (OSError, StopIteration):
        cpu_model = "unknown"
    out = {"value": B / medB, "unit": "graph-pairs/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
           "sample": f"batch {B}: {wB} warm-up + {nB} timed iterations (median; BASELINE.md section 2's 5 + 20); "
                     f"torch-CPU fp32, {cores} threads, reference op schedule materialising (B,E,D,D); restatement, "
                     f"not TensorFlow itself",
           "batch32": {"value": 32 / med32, "unit": "graph-pairs/s", "ms_per_batch": med32 * 1e3,
                       "sample": f"batch 32: {w32} warm-up + {n32} timed iterations (median)"},
           f"batch{B}": {"value": B / medB, "unit": "graph-pairs/s", "ms_per_batch": medB * 1e3}}
    if acc:
        out["gpu_vs_port_fp64"] = acc
    return out


def _gpu_timed(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def prepared_weights_cost(m, mode, iters=10):
    """What MPNNModel._prepared_weights keeps between forward calls, and what building it costs: the per-bond-type
    matrices A[v] = sum_k bond_table[v,k] W[k] of every step (the reference's per-forward Embedding + tensordot,
    models/layers.py:108) and the GatedUpdate kernels in MFMA operand order, rebuilt once per weight VERSION
    (load_weights / an optimizer step), not per forward.  -> microseconds per rebuild (both ions, all steps)."""
    def rebuild():
        m._prepared.pop(mode, None)
        m._prepared_weights(mode)
    return _gpu_timed(rebuild, iters) * 1e3


def time_other_configs(dev, Va, Vb):
    """Labelled extras of the bench line (never `value`): whole-model forward of BASELINE.json configs[2]
    (melting point, K = D^2 = 1024, S = 4, batch 8192: train_melting_point.py:146-198), of configs[4]'s forward
    shape (atom_dim 128, 6 steps, batch 4096: the validation / predict path of train_viscosity.py with atom_dim=128),
    configs[4]'s TRAINING step (forward + backward + Adam(clipnorm), train_viscosity.py:227-230,328-338) at batch 32
    and 4096, and the explicit-hydrogen padded shape N = 160, E = 640 of the real data sets; one stream, synthetic
    graphs, with the executed exact-f32 flops beside the time."""
    from ionic_mpnn_amd import model, synthetic, train, weights
    out = {}

    for name, B, D, K, S, kind in (("config3_melting_point_D32_K1024_S4_B8192", 8192, 32, 1024, 4, "melting_point"),
                                   ("config5_forward_D128_K8_S6_B4096", 4096, 128, 8, 6, "viscosity")):
        inp = synthetic.make_batch(B, seed=0, with_temperature=(kind == "viscosity"))
        if kind == "melting_point":
            m = model.build_melting_point_model(Va, Vb, atom_dim=D, num_steps=S, device=dev)
        else:
            m = model.build_model(Va, Vb, atom_dim=D, bond_dim=K, num_steps=S, device=dev)
        m.load_weights(weights.init_weights(kind, Va, Vb, atom_dim=D, bond_dim=K, num_steps=S, seed=1))
        d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
        ms = _gpu_timed(lambda: m(d), 10)
        rows, edges = executed_counts(inp)
        fl = S * (12 * D * D * rows + 2 * D * D * edges)
        mode = m.resolve_encoder_mode(inp["cat_atom"].shape[1], inp["cat_bond"].shape[1])
        modes_timed = {mode: ms}
        if mode == "f32t":
            # the same forward with GEMMs as f32 (bf16x9 emulation) - D = 32: the GatedUpdate GEMMs; D = 128: those of the
            # message layers too.  The judge's ruling as for the headline: both modes timed in this run, the faster one
            # carries the entry's figure, the exact-f32 figure stays beside it; tests/test_gpu_encoder.py and
            # tests/test_gpu_wide.py hold the error / magnitude / NaN / bitwise conditions
            m.encoder_mode = "f32x3"
            ms3 = _gpu_timed(lambda: m(d), 10)
            modes_timed["f32x3"] = ms3
            if ms3 < ms:
                ms, mode = ms3, "f32x3"
            else:
                m.encoder_mode = "auto"
        prep_us = prepared_weights_cost(m, mode, 5)

        def with_prepare():
            m._prepared.pop(mode, None)
            return m(d)
        ms_prep = _gpu_timed(with_prepare, 10)
        out[name] = {"ms_per_forward": ms, "graph_pairs_per_s": B / (ms * 1e-3), "encoder": mode,
                     "modes_timed_ms_per_forward": modes_timed,
                     "dtype": "f32 (bf16x9 emulation)" if mode == "f32x3" else "f32",
                     "executed_f32_tflops": fl / (ms * 1e-3) / 1e12,
                     "frac_of_f32_mfma_peak": fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                     "prepared_weights_us_per_weight_version": prep_us,
                     "ms_per_forward_with_prepare": ms_prep,
                     "graph_pairs_per_s_with_prepare": B / (ms_prep * 1e-3),
                     "note": "whole model forward incl. plan kernels and the head, one stream; ms_per_forward: weights frozen "
                             "across calls (inference: the per-bond-type matrices A[v] = sum_k Tb[v,k] W[k] and the operand-order "
                             "GatedUpdate kernels are kept per weight version); ms_per_forward_with_prepare: rebuilt in every "
                             "call, as the reference recomputes its tensordot per forward (models/layers.py:108); executed flops "
                             "= exact-f32 products on kept rows / valid edges (12 D^2 per row, 2 D^2 per edge and step)"}
        del m, d
        torch.cuda.empty_cache()

    # explicit-hydrogen padded shape of the real data sets (src/featurize.py:45, train_viscosity.py:288-289): bond ids
    # uniform over the 71-type stand-in vocabulary (as every other synthetic batch of this file), and once more with the
    # type statistics real molecules have - a handful of bond types with skewed frequencies, so that the typed encoder's
    # groups of 4 edges per type are full
    B, N, E, S = 4096, 160, 640, 3
    for name, probs in (("explicit_h_shape_N160_E640_D32_K8_S3_B4096", None),
                        ("explicit_h_shape_6_bond_types_N160_E640_D32_K8_S3_B4096", (0.55, 0.2, 0.12, 0.07, 0.04, 0.02))):
        inp = synthetic.make_explicit_h_batch(B, max_atoms=N, max_edges=E, seed=0, bond_type_probs=probs)
        m = model.build_model(Va, Vb, atom_dim=32, bond_dim=8, num_steps=S, device=dev)
        m.load_weights(weights.init_weights("viscosity", Va, Vb, atom_dim=32, bond_dim=8, num_steps=S, seed=1))
        d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
        modes = {}
        for mode in ("f32t", "f32x3"):
            m.encoder_mode = mode
            modes[mode] = _gpu_timed(lambda: m.encode_pooled(d), 10)
        best = min(modes, key=modes.get)
        ms = modes[best]
        rows, edges = executed_counts(inp)
        fl = S * (12 * 32 * 32 * rows + 2 * 32 * 32 * edges)
        out[name] = {
            "ms_per_encode": ms, "graph_pairs_per_s": B / (ms * 1e-3), "encoder": best, "modes_timed_ms_per_encode": modes,
            "dtype": "f32 (bf16x9 emulation)" if best == "f32x3" else "f32",
            "overflow_fallbacks": int(getattr(m, "overflow_fallbacks", 0)), "kept_rows": rows, "valid_edges": edges,
            "executed_f32_tflops": fl / (ms * 1e-3) / 1e12,
            "note": "encode() of both ions (plan + fused typed encoder, 640-edge chunks; one 4-byte read-back of the plan's "
                    "overflow word per call); molecules of 20-160 atoms, degree <= 4, every bond in four edge slots; bond ids "
                    + ("uniform over 71 types" if probs is None else "from 6 types with frequencies %s" % (probs,))}
        del m, d
        torch.cuda.empty_cache()

    # the same padded shape at config 5's model size (atom_dim 128, 6 steps) through the wide encoder (E <= 1024 there)
    B, N, E, D, S = 1024, 160, 640, 128, 6
    inp = synthetic.make_explicit_h_batch(B, max_atoms=N, max_edges=E, seed=0)
    m = model.build_model(Va, Vb, atom_dim=D, bond_dim=8, num_steps=S, device=dev)
    m.load_weights(weights.init_weights("viscosity", Va, Vb, atom_dim=D, bond_dim=8, num_steps=S, seed=1))
    d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    modes = {}
    for mode in ("f32t", "f32x3"):
        m.encoder_mode = mode
        modes[mode] = _gpu_timed(lambda: m.encode_pooled(d), 6)
    best = min(modes, key=modes.get)
    rows, edges = executed_counts(inp)
    fl = S * (12 * D * D * rows + 2 * D * D * edges)
    out["explicit_h_shape_N160_E640_D128_K8_S6_B1024"] = {
        "ms_per_encode": modes[best], "graph_pairs_per_s": B / (modes[best] * 1e-3), "encoder": best,
        "modes_timed_ms_per_encode": modes, "dtype": "f32 (bf16x9 emulation)" if best == "f32x3" else "f32",
        "kept_rows": rows, "valid_edges": edges, "executed_f32_tflops": fl / (modes[best] * 1e-3) / 1e12,
        "frac_of_f32_mfma_peak": fl / (modes[best] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
        "note": "encode() of both ions through the wide encoder (plan + run in one call) at the explicit-hydrogen padded "
                "shape; executed flops = exact-f32 products on kept rows / valid edges"}
    del m, d
    torch.cuda.empty_cache()

    # configs[4]: the full training step at atom_dim 128, 6 steps
    D, K, S = 128, 8, 6
    # (eager launches at batch 4096: 15.8 vs 15.2 ms); third entry: the reference's batch at the padded shape of its real
    # (explicit-hydrogen) data sets, N = 160, E = 640
    for B, graphed, iters, explicit_h in ((32, True, 100, False), (4096, True, 8, False), (32, True, 100, True)):
        inp = synthetic.make_explicit_h_batch(B, seed=0) if explicit_h else synthetic.make_batch(B, seed=0)
        y = np.random.default_rng(0).normal(4.0, 1.0, size=B).astype(np.float32)
        m = model.build_model(Va, Vb, atom_dim=D, bond_dim=K, num_steps=S, device=dev)
        m.load_weights(weights.init_weights("viscosity", Va, Vb, atom_dim=D, bond_dim=K, num_steps=S, seed=1))
        m.compile(train.Adam(1e-3, clipnorm=1.0))
        d = m._to_device(inp)
        step = train.GraphedTrainStep(m, d, y) if graphed else m.train_on_batch
        ms = _gpu_timed(lambda: step(d, y), iters, warm=3)
        rows, edges = executed_counts(inp)
        fwd = S * (12 * D * D * rows + 2 * D * D * edges)
        out[("explicit_h_train_step_N160_E640_D128_K8_S6_B%d" if explicit_h else "config5_train_step_D128_K8_S6_B%d") % B] = {
            "ms_per_train_step": ms, "graph_pairs_per_s": B / (ms * 1e-3), "hipgraph": graphed,
            "executed_f32_tflops": 3 * fwd / (ms * 1e-3) / 1e12,
            "frac_of_f32_mfma_peak": 3 * fwd / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
            "note": "forward + backward + Adam(1e-3, clipnorm=1.0) + MSE + l2 (train_viscosity.py:189,227-230,328-338), one "
                    "GPU, " + ("one captured hipGraph replayed per step" if graphed else "eager launches") +
                    "; executed flops counted as 3 x the forward's exact-f32 products (forward, data gradient, weight gradient)"}
        del m, d, step
        torch.cuda.empty_cache()
    return out


def run_config5(args, rank, world, dev, rehearsal, ndev):
    """--config5: BASELINE.json configs[4] as worded - the full train_viscosity.py step (forward + backward + Adam(1e-3,
    clipnorm=1.0) + MSE + l2; train_viscosity.py:189,227-230,328-338) at atom_dim 128, 6 message-passing steps, data-parallel
    over the ranks: every timed step is MPNNModel.train_on_batch on the rank's shard, i.e. it ends with the all-reduce of
    the flat gradient buffer (RCCL) before the optimizer step.  Eager launches at every N (a captured graph would have to
    hold the collective); the single-GPU graphed figure is `other_configs.config5_train_step_*` of the default line."""
    import torch.distributed as dist
    from ionic_mpnn_amd import model, synthetic, train, weights
    Va, Vb, D, K, S = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, 128, 8, 6
    B = args.batch if args.batch is not None else 4096
    inp = synthetic.make_batch(B, seed=rank)
    y = np.random.default_rng(rank).normal(4.0, 1.0, size=B).astype(np.float32)
    m = model.build_model(Va, Vb, atom_dim=D, bond_dim=K, num_steps=S, device=dev)
    m.load_weights(weights.init_weights("viscosity", Va, Vb, atom_dim=D, bond_dim=K, num_steps=S, seed=1))
    m.compile(train.Adam(1e-3, clipnorm=1.0))
    d = m._to_device(inp)
    step = lambda: m.train_on_batch(d, y, n_global=B * world)
    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    loss = float(loss)
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    rows, edges = executed_counts(inp)
    fwd = S * (12 * D * D * rows + 2 * D * D * edges)
    ms = elapsed / args.steps * 1e3
    tfl = 3 * fwd / (ms * 1e-3) / 1e12
    params = int(sum(t.numel() for _, t in m.trainable_variables()))
    out = {
        "metric": "molecule-graph pairs/sec (training step: fwd + bwd + Adam), per MI355X batch %d" % B,
        "value": B * world * args.steps / elapsed, "unit": "graph-pairs/s", "n_gpus": world, "steps": args.steps,
        "warmup": max(args.warmup, 1), "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BASELINE.json configs[4]: full train_viscosity.py step (forward + backward + Adam(1e-3, "
                               "clipnorm=1.0) + MSE + l2), atom_dim 128, bond_dim 8, 6 message-passing steps, synthetic padded "
                               "graphs N<=40 E<=80, batch %d pairs/GPU, eager launches" % B,
                   "parallelism": "data-parallel x%d: batch-sharded, weights replicated, one all-reduce of the flat gradient "
                                  "buffer (%d floats) per step before the optimizer step" % (world, params),
                   "global_batch": B * world, "loss_last": loss, "trainable_parameters": params},
        "roofline": {"bound": "mfma", "achieved": tfl, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": tfl / PEAK_F32_MFMA_TFLOPS, "traffic": None,
                     "note": "this rank's executed exact-f32 products counted as 3 x the forward's (forward, data gradient, "
                             "weight gradient) over the whole step; per-kernel figures: profiles/r3_train_config5_*"},
        "cpu_baseline": None,
    }
    if rehearsal:
        out["config"]["rehearsal"] = f"{world} ranks share {ndev} GPU(s) over gloo - not a scaling number"
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=None, help="graph pairs per GPU (default 4096; --config4: 8192)")
    ap.add_argument("--mp-steps", type=int, default=3)
    ap.add_argument("--config4", action="store_true",
                    help="BASELINE.json configs[3] as worded: 8192 pairs per rank, every timed step ends with the all-gather of "
                         "the pooled fingerprints and the all-reduce of the loss statistics")
    ap.add_argument("--config5", action="store_true",
                    help="BASELINE.json configs[4] as worded: the full training step (fwd + bwd + Adam) at atom_dim 128, 6 steps, "
                         "data-parallel over the ranks with the gradient all-reduce inside every timed step (default batch 4096 "
                         "pairs per rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the labelled extras for BASELINE.json configs[2] and configs[4]")
    ap.add_argument("--schedule", choices=["fused", "layered"], default="fused")
    ap.add_argument("--pipeline", action="store_true",
                    help="enqueue the plan kernels of step i+1 on a side stream before the encoder of step i "
                         "(default: plan and encode every batch back to back on one stream, which is faster on "
                         "MI355X: the encoder fills every CU's register file, see include/impnn.h)")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams that take consecutive batches in turn (fused schedule; 1: every launch on one "
                         "stream): the plan kernels and the uneven tail of one batch's persistent encoder overlap "
                         "the neighbouring batches' kernels")
    ap.add_argument("--encoder-workgroups", type=int, default=None,
                    help="persistent workgroups per encoder launch (the `workgroups` argument of the encoder entries; 0 = one per CU). "
                         "Default: 128 with 3 or more streams, else 0")
    ap.add_argument("--ramp-ms", type=float, default=150.0,
                    help="untimed clock ramp before the W warm-up steps: the same step() repeated for this many "
                         "milliseconds (0 disables)")
    ap.add_argument("--mode", choices=["auto", "f32t", "f32x3", "f32", "f16x2"], default="auto",
                    help="schedule / arithmetic of the fused encoder (include/impnn.h).  auto: the exact-f32 form f32t AND its "
                         "bf16x9 emulation f32x3 are both timed over the same K steps; `value` is f32x3's only where it is the "
                         "faster of the two in this run (VERDICT r2, ruling on mode f32x3), else f32t's.  f16x2 (narrower "
                         "products) on request only")
    args = ap.parse_args()

    from ionic_mpnn_amd import _lib, dist as idist, model, ops, synthetic, weights

    rank, local_rank, world = idist.env_world()
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 through torch.distributed.run (see module docstring)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    rehearsal = world > ndev  # more ranks than GPUs (1-GPU box rehearsal): share devices, gloo
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        idist.init_distributed(backend="gloo" if rehearsal else "nccl")
    import torch.distributed as dist

    if args.config5:
        return run_config5(args, rank, world, dev, rehearsal, ndev)

    N, E, D, K, S = 40, 80, 32, 8, args.mp_steps
    B = args.batch if args.batch is not None else (8192 if args.config4 else 4096)
    # every rank draws its own shard of the global batch (seed offset by rank); weights replicated
    inputs = synthetic.make_batch(B, max_atoms=N, max_edges=E, seed=0 + rank)
    w = weights.init_weights("viscosity", synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, atom_dim=D, bond_dim=K,
                             num_steps=S, seed=1)
    m = model.build_model(synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, atom_dim=D, bond_dim=K, num_steps=S, device=dev)
    m.load_weights(w)
    d_in = {k: torch.from_numpy(v).to(dev) for k, v in inputs.items()}  # resident in HBM before timing
    fused = args.schedule == "fused"
    pipelined = fused and args.pipeline and S > 0

    # --streams n: consecutive batches go to n HIP streams in turn, so the plan kernels and the uneven tail of one
    # batch's persistent encoder overlap the next batch's kernels; every step still plans and encodes one full batch
    enc_wgs = args.encoder_workgroups
    if enc_wgs is None:
        enc_wgs = 128 if (args.streams >= 3 and fused and not args.pipeline and not args.config4) else 0
    n_lanes = args.streams if (args.streams > 1 and fused and not pipelined and not args.config4) else 0
    lanes = [torch.cuda.Stream(device=dev) for _ in range(n_lanes)]
    for ln in lanes:
        ln.wait_stream(torch.cuda.current_stream(dev))
    lib = _lib.load()

    # ---- BASELINE configs[3]'s collectives: all-gather of the per-rank fingerprints (per-sample rows: the semantically
    # correct "all-reduce of pooled fingerprints") + all-reduce of the two loss scalars, once per step
    y_target = torch.zeros(B, dtype=torch.float32, device=dev)
    gathered = {"fp": None}

    def collectives(pc, pa):
        fp = torch.cat([pc, pa], dim=1)  # (B, 2 D) fingerprints of this rank's shard
        if world == 1:
            gathered["fp"] = fp
            return idist.all_reduce_loss_stats(fp.sum(dim=1), y_target)
        if rehearsal:  # gloo on host copies (ranks share a GPU): a rehearsal of the code path, not a number
            out = [torch.empty(fp.shape, dtype=fp.dtype) for _ in range(world)]
            dist.all_gather(out, fp.cpu())
            gathered["fp"] = torch.cat(out)
            diff = (fp.sum(dim=1) - y_target).double().cpu()
            stats = torch.stack([(diff * diff).sum(), torch.tensor(float(diff.numel()), dtype=torch.float64)])
            dist.all_reduce(stats)
            return stats
        out = torch.empty((world * B, fp.shape[1]), dtype=fp.dtype, device=dev)
        dist.all_gather_into_tensor(out, fp)  # equal shards: one RCCL all-gather, no size exchange
        gathered["fp"] = out
        return idist.all_reduce_loss_stats(fp.sum(dim=1), y_target)

    def run_mode(mode, with_collectives, ramp_ms):
        """ramp + W warm-up + EXACTLY K timed steps of `mode`, bracketed by barrier + synchronize on both sides
        -> (max-over-ranks elapsed seconds, last outputs, untimed ramp steps)."""
        m.encoder_mode = mode
        m.encoder_workgroups = enc_wgs
        state = {"plan": m.plan_batch(d_in) if pipelined else None, "i": 0}

        def step():
            if lanes:
                ln = lanes[state["i"] % len(lanes)]
                state["i"] += 1
                with torch.cuda.stream(ln):
                    return m.encode_pooled(d_in, fused=fused)
            if not pipelined:
                out = m.encode_pooled(d_in, fused=fused)
            else:
                nxt = m.plan_batch(d_in)  # the next step's batch (same synthetic graphs, planned again from scratch)
                out = m.encode_pooled(d_in, plan=state["plan"])
                state["plan"] = nxt
            if with_collectives:
                collectives(*out)
            return out

        # Untimed: bring the GPU to its sustained clock first.  An idle MI355X runs the first few hundred
        # launches ~13 % slower (measured: 92 us -> 80 us per encoder launch after ~30 ms of load), and the
        # default W=10 warm-up is 1 ms of work.  Same step() as the timed loop.
        ramp_steps = 0
        t_ramp = time.perf_counter()
        while (time.perf_counter() - t_ramp) * 1e3 < ramp_ms:
            for _ in range(20):
                pc, pa = step()
            torch.cuda.synchronize()
            ramp_steps += 20
        for _ in range(args.warmup):
            pc, pa = step()
        torch.cuda.synchronize()
        if fused:
            _lib.check(lib.impnn_profile_enable(args.steps))
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pc, pa = step()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0  # this rank's K steps; the job time is the MAX over ranks
        if world > 1:
            dist.barrier()
            tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            elapsed = float(tmax.item())
        overlapped_ms = None
        if fused:
            buf = (C.c_float * args.steps)()
            n = C.c_int32(0)
            _lib.check(lib.impnn_profile_collect(buf, args.steps, C.byref(n)))
            lib.impnn_profile_disable()
            if n.value:
                overlapped_ms = float(np.mean(np.frombuffer(buf, dtype=np.float32, count=n.value)))
        return {"elapsed": elapsed, "pc": pc, "pa": pa, "ramp_steps": ramp_steps, "overlapped_ms": overlapped_ms,
                "step": step}

    # ---- the timed region(s) ---------------------------------------------------------------------------------------
    if not fused:
        candidates = ["layered"]
    elif args.mode == "auto":
        m.encoder_mode = "auto"
        first = m.resolve_encoder_mode(N, E)
        candidates = [first]
        m.encoder_mode = "f32x3"
        if first == "f32t" and m.resolve_encoder_mode(N, E) == "f32x3":
            candidates.append("f32x3")
    else:
        m.encoder_mode = args.mode
        candidates = [m.resolve_encoder_mode(N, E)]
    runs = {}
    for i, mode in enumerate(candidates):
        runs[mode] = run_mode(mode if fused else "auto", args.config4, args.ramp_ms if i == 0 else min(args.ramp_ms, 30.0))
    # VERDICT r2, ruling on f32x3: it may carry `value` only where it beats f32t in the timed configuration of this run
    mode_used = min(runs, key=lambda k: runs[k]["elapsed"]) if len(runs) > 1 else candidates[0]
    main_run = runs[mode_used]
    elapsed, pc, pa = main_run["elapsed"], main_run["pc"], main_run["pa"]
    no_coll = None
    if args.config4:  # the same K steps without the collectives, beside it
        no_coll = run_mode(mode_used, False, 30.0)["elapsed"]

    # epilogue collective (outside the per-sample data path): global fingerprint checksum
    local_sum = torch.stack([pc.double().sum() + pa.double().sum(),
                             torch.tensor(float(B), dtype=torch.float64, device=dev)])
    if world > 1:
        t = local_sum.cpu() if rehearsal else local_sum
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        local_sum = t
    total_pairs = float(local_sum[1].item())

    # ---- everything below is outside the timed region ----------------------------------------------------------
    def events_ms(fn, n):
        """mean HIP-event duration (ms) of the encoder kernel over n calls of fn on the current stream: the events
        are recorded by libimpnn around that launch alone, on the stream it is launched on"""
        _lib.check(lib.impnn_profile_enable(n))
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        buf = (C.c_float * n)()
        cnt = C.c_int32(0)
        _lib.check(lib.impnn_profile_collect(buf, n, C.byref(cnt)))
        lib.impnn_profile_disable()
        return float(np.mean(np.frombuffer(buf, dtype=np.float32, count=cnt.value))) if cnt.value else None

    def kernel_alone_ms(mode):
        """K back-to-back launches of the encoder kernel alone (one plan, impnn_encoder_run K times) between two HIP
        events on the launch stream: (stop - start) / K.  An event pair around ONE launch also times the
        dispatch / scratch set-up / end-of-kernel release on either side of it (measured: ~18 us on a 140 us
        kernel, against rocprofv3's begin/end timestamps of the same run); per-launch pairs stay in `overlapped`."""
        m.encoder_mode = mode
        m.encoder_workgroups = 0
        for _ in range(5):
            m.encode_pooled(d_in, fused=True)
        if S == 0:
            return events_ms(lambda: m.encode_pooled(d_in, fused=True), args.steps)
        plan = m.plan_batch(d_in)
        torch.cuda.current_stream(dev).wait_event(plan.ready)
        prep = m._prepared_weights(plan.mode)
        pooled = [torch.empty(B, D, dtype=torch.float32, device=dev) for _ in range(2)]
        arr = C.c_void_p * 2
        ids_p = arr(*[p_[0].data_ptr() for p_ in plan.ions])
        prep_p, pool_p = arr(*[t.data_ptr() for t in prep]), arr(*[t.data_ptr() for t in pooled])
        at, bt = m.atom_emb.embeddings, m.bond_emb.embeddings
        ws = plan.slot["ws"]
        strm = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)

        def run_once():  # the bare C call: ~10 us of host time, so K launches queue back to back on the stream
            _lib.check(lib.impnn_encoder_run(2, ids_p, C.c_void_p(at.data_ptr()), at.shape[0], C.c_void_p(bt.data_ptr()),
                                             bt.shape[0], prep_p, ops.ENCODER_MODES[plan.mode], pool_p, B, N, E, D, K, S,
                                             C.c_float(ops.LN_EPS), C.byref(plan.info), C.c_void_p(ws.data_ptr()),
                                             ws.numel(), strm))
        for _ in range(3):
            run_once()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.steps):
            run_once()
        e1.record()
        torch.cuda.synchronize()
        assert torch.equal(pooled[0], m.encode_pooled(d_in, fused=True)[0])
        return e0.elapsed_time(e1) / args.steps

    def single_stream_ms(mode):
        m.encoder_mode = mode
        m.encoder_workgroups = 0
        return _gpu_timed(lambda: m.encode_pooled(d_in, fused=True), args.steps, warm=5)

    kernel_ms, single_ms, full_ms, extras, other_configs, prep_us = {}, {}, None, {}, None, None
    if fused and rank == 0:
        for mode in runs:
            kernel_ms[mode] = kernel_alone_ms(mode)
            single_ms[mode] = single_stream_ms(mode)
        if world == 1:
            m.encoder_mode = mode_used
            m.encoder_workgroups = 0
            full_ms = _gpu_timed(lambda: m(d_in, fused=True), args.steps)
            prep_us = prepared_weights_cost(m, mode_used)
            # labelled extras: the other encoder modes on the same batch, single stream, with their own error against
            # the timed mode's result (never `value`: "f16x2" is narrower arithmetic than the reference's f32)
            ref_c, ref_a = m.encode_pooled(d_in, fused=True)
            scale = float(max(ref_c.abs().max(), ref_a.abs().max()))
            for other in ("f32t", "f32x3", "f32", "f16x2"):
                if other == mode_used:
                    continue
                m.encoder_mode = other
                if m.resolve_encoder_mode(N, E) != other:
                    continue
                oc, oa = m.encode_pooled(d_in, fused=True)
                k_ms = kernel_ms.get(other) or events_ms(lambda: m.encode_pooled(d_in, fused=True), args.steps)
                s_ms = single_ms.get(other) or single_stream_ms(other)
                extras[other] = {"kernel_ms": k_ms, "ms_per_step_single_stream": s_ms,
                                 "graph_pairs_per_s_single_stream": B / (s_ms * 1e-3),
                                 "max_rel_diff_vs_timed_mode": float(max((oc - ref_c).abs().max(),
                                                                         (oa - ref_a).abs().max())) / scale}
                if other in runs:  # timed over the same K steps in the configuration of `value`
                    extras[other]["graph_pairs_per_s_timed_configuration"] = total_pairs * args.steps / runs[other]["elapsed"]
                elif lanes:
                    m.encoder_workgroups = enc_wgs
                    state = {"i": 0}

                    def lane_step():
                        ln = lanes[state["i"] % len(lanes)]
                        state["i"] += 1
                        with torch.cuda.stream(ln):
                            m.encode_pooled(d_in, fused=True)
                    l_ms = _gpu_timed(lane_step, args.steps, warm=3 * len(lanes))
                    extras[other][f"graph_pairs_per_s_{len(lanes)}_streams"] = B / (l_ms * 1e-3)
                    m.encoder_workgroups = 0
            m.encoder_mode = mode_used
            if not args.no_other_configs:
                other_configs = time_other_configs(dev, synthetic.DEFAULT_VA, synthetic.DEFAULT_VB)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = total_pairs * args.steps / elapsed
    flops_launch = algorithmic_flops_per_pair(N, E, D, S) * B
    bytes_launch = algorithmic_bytes_per_pair(N, E, D) * B
    kept_rows, valid_edges = executed_counts(inputs)
    typed = mode_used in ("f32t", "f32x3")
    msg_flops = S * (2 * D * D * valid_edges if typed else 2 * K * D * D * kept_rows)
    upd_flops = S * 12 * D * D * kept_rows
    executed_flops = upd_flops + msg_flops
    arith = {"f16x2": "f32 in/out/accumulate; every f32 GEMM product formed from fp16 hi/lo splits (3 "
                      "v_mfma_f32_16x16x32_f16 per f32 product, product error ~2^-21: NARROWER than the reference's f32)",
             "f32": "exact f32 products on v_mfma_f32_16x16x4_f32 (pull form: agg = sum_k W_k G_k)",
             "f32t": "exact f32 products: per-bond-type messages (models/layers.py:108-112 in the reference's own order) "
                     "on v_mfma_f32_4x4x1_16b_f32, GatedUpdate on v_mfma_f32_16x16x4_f32; f32 accumulate, f32 in/out",
             "f32x3": "as f32t, with the GatedUpdate GEMMs on the bf16 matrix pipe: every f32 operand carried exactly as three "
                      "bf16 terms, all nine cross products accumulated in f32 (9 v_mfma_f32_16x16x32_bf16 per 8 f32 MFMAs) - "
                      "the f32 products themselves, summed in another order (VERDICT r2: admissible as 'f32 (bf16x9 "
                      "emulation)' where it beats f32t in the timed configuration; the f32t figure stays in this line)",
             "layered": "f32, one launch per reference layer"}[mode_used]
    dtype = {"f16x2": "f16x2 (split-fp16 products, f32 accumulate)", "f32x3": "f32 (bf16x9 emulation)"}.get(mode_used, "f32")
    workload = (f"BASELINE.json configs[{3 if args.config4 else 1}]: message-passing forward (embedding gather -> {S}x"
                f"(BondMatrixMessage, Reduce, GatedUpdate) -> GlobalSumPool), cation+anion, synthetic padded graphs N<={N} "
                f"E<={E}, D={D}, K={K}, batch {B} pairs/GPU, schedule={args.schedule}")
    out = {
        "metric": "molecule-graph pairs/sec (fwd), batch 4096 per MI355X",  # BASELINE.json's metric; 1 pair = 2 graphs
        "value": value, "unit": "graph-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype, "data": "synthetic",
        "config": {"workload": workload, "arithmetic": arith, "mode": mode_used,
                   "modes_timed": {k: {"graph_pairs_per_s": total_pairs * args.steps / r["elapsed"],
                                       "ms_per_step": r["elapsed"] / args.steps * 1e3} for k, r in runs.items()},
                   "mode_rule": ("both exact-f32 forms were timed over the same K steps in the same configuration (own ramp + "
                                 "W warm-up each); `value` is the faster one, f32x3 only when it beats f32t (VERDICT r2)")
                   if len(runs) > 1 else "one mode timed",
                   "pipeline": ("plan kernels of step i+1 run on a side stream under the encoder of step i; every "
                                "step still plans and encodes one full batch") if pipelined else "none",
                   "streams": (f"{len(lanes)} HIP streams take consecutive batches in turn (each batch: plan + encoder in "
                               "stream order on its own workspace); kernels of neighbouring batches overlap; "
                               f"{enc_wgs or 'one per CU:'} persistent workgroups per encoder launch") if lanes
                   else "1 (every launch on one stream)",
                   "resident_batch": "every step plans and encodes the SAME resident batch from scratch: no result, plan or "
                                     "chunk record is carried from one step to the next; its 10 MB of inputs stay warm in "
                                     "L2/MALL - immaterial here: the path is compute-bound (HBM < 1 % of 8 TB/s)",
                   "prepared_weights": {
                       "what": "kept between steps per WEIGHT VERSION (not per batch): the per-bond-type matrices A[v] = sum_k "
                               "bond_table[v,k] W[k] of all 2 S message layers - the reference's per-forward Embedding + "
                               "tensordot (models/layers.py:108) - and the GatedUpdate kernels in MFMA operand order "
                               "(MPNNModel._prepared_weights: rebuilt after load_weights / an optimizer step; frozen-weight "
                               "inference therefore pays it once)",
                       "us_per_weight_version": prep_us,
                       "share_of_a_step_if_rebuilt_every_forward": (prep_us * 1e-3 / ms_per_step) if prep_us else None},
                   "global_batch": int(total_pairs), "molecule_graphs_per_s": 2.0 * value,
                   "parallelism": f"batch-sharded x{world}, weights replicated, " +
                                  ("every timed step ends with the RCCL all-gather of the (B, 2 D) fingerprints and the "
                                   "all-reduce of the two loss scalars" if args.config4 else
                                   "no data-path collective; one all-reduce of the fingerprint checksum after the timed region"),
                   "clock_ramp": f"{main_run['ramp_steps']} untimed steps before the {args.warmup} warm-up steps, "
                                 "so that the timed steps run at the sustained GPU clock",
                   "checksum": float(local_sum[0].item())},
    }
    if args.config4:
        out["config"]["collective_us"] = (elapsed - no_coll) / args.steps * 1e6
        out["config"]["without_collectives"] = {"graph_pairs_per_s": total_pairs * args.steps / no_coll,
                                                "ms_per_step": no_coll / args.steps * 1e3}
        out["config"]["gathered_fingerprint_rows"] = int(gathered["fp"].shape[0]) if gathered["fp"] is not None else None
    if single_ms.get(mode_used):
        s_ms = single_ms[mode_used]
        out["config"]["single_stream"] = {"ms_per_step": s_ms, "graph_pairs_per_s": B / (s_ms * 1e-3),
                                          "note": "plan + encoder of one batch at a time on ONE stream, one workgroup "
                                                  "per CU (untimed extra steps of this run)"}
    if full_ms:
        out["config"]["full_model_forward"] = {"ms_per_step": full_ms, "graph_pairs_per_s": B / (full_ms * 1e-3),
                                               "note": "hot path + heads (impnn_model_head) -> log_eta, single stream; "
                                                       "not `value`"}
    if extras:
        out["config"]["other_modes"] = extras
    if other_configs:
        out["config"]["other_configs"] = other_configs
    if rehearsal:
        out["config"]["rehearsal"] = f"{world} ranks share {ndev} GPU(s) over gloo - not a scaling number"

    def roofline_of(mode):
        k_ms = kernel_ms.get(mode) or runs[mode]["overlapped_ms"]
        if not k_ms:
            return None
        kname = "encoder_typed_kernel" if mode in ("f32t", "f32x3") else "encoder_fused_kernel"
        pkey = "encoder_typed_x3" if mode == "f32x3" else kname
        traffic, tsrc = pmc_field("hbm_bytes_per_launch", pkey)
        busy, _ = pmc_field("mfma_pipe_busy_frac", pkey)
        ach = flops_launch / (k_ms * 1e-3) / 1e12
        r = {"bound": "mfma", "kernel": kname + ("<x3>" if mode == "f32x3" else ""), "achieved": ach,
             "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS, "traffic": traffic,
             "kernel_ms": k_ms,
             "traffic_source": f"profiles/{tsrc}: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch from separate rocprofv3 --pmc "
                               "passes of this command (tools/pmc_profile.sh) on an earlier box run of the same kernel - "
                               "PMC collection cannot share a run with the timing",
             "timing": f"two HIP events around {args.steps} back-to-back launches of the kernel alone on the chip "
                       "(one plan, impnn_encoder_run K times, one stream, one workgroup per CU; untimed "
                       "extra launches of this run), divided by K",
             "note": "achieved = SURVEY 8(d) algorithmic f32 flops (2 S (2 D^2 E + 12 D^2 N) per pair) per launch / kernel_ms "
                     "against the dense f32 MFMA peak; the algorithmic count includes padding atoms and padding edge slots, "
                     "which the kernel skips exactly: `executed_flops_per_launch` is what it multiplies",
             "algorithmic_flops_per_launch": flops_launch,
             "executed_flops_per_launch": executed_flops,
             "executed_achieved": executed_flops / (k_ms * 1e-3) / 1e12,
             "executed_frac_of_f32_peak": executed_flops / (k_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
             "matrix_pipe_busy_frac_pmc": busy,
             "hbm": {"algorithmic_bytes_per_launch": bytes_launch,
                     "achieved_GBs": bytes_launch / (k_ms * 1e-3) / 1e9,
                     "frac_of_8TBs": bytes_launch / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}}
        if mode == "f32x3":
            # the GatedUpdate products run as 9 bf16 MFMA products each on the bf16 pipe, the messages on the f32 pipe:
            # time both pipes would need at their peaks for the EXECUTED work / the kernel's time
            t_peak = msg_flops / (PEAK_F32_MFMA_TFLOPS * 1e12) + 9.0 * upd_flops / (PEAK_BF16_MFMA_TFLOPS * 1e12)
            r["pipe_time_at_peak_frac"] = t_peak / (k_ms * 1e-3)
            r["pipe_note"] = ("f32 (bf16x9 emulation): executed message flops / 157.3 TFLOP/s (f32 MFMA) + 9 x executed "
                              "GatedUpdate flops / 2500 TFLOP/s (bf16 MFMA), over kernel_ms - the two pipes' time at peak; "
                              "`frac` above stays the f32-equivalent figure against the f32 peak (can exceed what an "
                              "exact-f32 kernel may reach: 9/16 of the f32 MFMA time per product)")
        ov = runs[mode]["overlapped_ms"]
        if lanes and ov:
            r["overlapped"] = {"streams": len(lanes), "kernel_ms": ov,
                               "note": "per launch as HIP events / rocprofv3 see it inside the timed loop, while the "
                                       "neighbouring batches' kernels share the chip - not a roofline figure"}
        return r

    rf = roofline_of(mode_used) if fused else None
    if rf:
        rf["whole_step"] = {"achieved": flops_launch / (ms_per_step * 1e-3) / 1e12,
                            "frac": flops_launch / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                            "note": "algorithmic flops / ms_per_step (plan kernels included)"}
        out["roofline"] = rf
        for other in runs:
            if other != mode_used:
                out[f"roofline_{other}"] = roofline_of(other)
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(inputs, w, gpu_pooled=(pc, pa))
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
