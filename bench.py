#!/usr/bin/env python3
"""bench.py - graph-pairs/sec of the message-passing forward hot path on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 10
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path (embedding gather, S x (BondMatrixMessage,
Reduce, GatedUpdate), GlobalSumPool, for the cation AND the anion branch) over one resident batch
of synthetic padded graph pairs: BASELINE.json configs[1] (N=40, E=80, D=32, K=8, S=3, batch 4096
per GPU; weak scaling: every rank owns its own 4096 pairs, no data-path collective).
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

ROOT = Path(__file__).resolve().parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32 MFMA peak (16x16x4 / 4x4x1 f32: 256 flop/clk/CU)
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense fp16 MFMA peak
PEAK_HBM_GBS = 8000.0          # HBM3E spec


def algorithmic_flops_per_pair(N, E, D, S):
    """SURVEY.md 8(d), per-bond-type schedule: 2 ions * S * (2 D^2 E + 12 D^2 N)."""
    return 2 * S * (2 * D * D * E + 12 * D * D * N)


def algorithmic_bytes_per_pair(N, E, D):
    """SURVEY.md 8(d), fused forward: 2*(4N + 12E + 4D) (+8 for T and the output scalar)."""
    return 2 * (4 * N + 12 * E + 4 * D) + 8


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


def pmc_field(name, kernel="encoder_fused"):
    """A figure for the encoder kernel from the committed rocprofv3 --pmc passes (profiles/pmc_*.json,
    latest round wins), or None.  hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: the gfx950
    FETCH_SIZE correction of MI355X_MICROARCH.md's HBM section.  (PMC passes are separate rocprofv3 runs of this
    same command, tools/pmc_profile.sh; the figure is per launch of the named kernel.)"""
    best = None
    key = "encoder_typed" if "typed" in kernel else "encoder_fused"
    for f in sorted((ROOT / "profiles").glob("pmc_*.json")):
        try:
            best = json.loads(f.read_text()).get(key, {}).get(name, best)
        except (OSError, ValueError):
            pass
    return best


def _time_cpu(fn, warmup, iters, budget_s):
    """median seconds of fn() over up to `iters` timed runs after `warmup` untimed ones, inside a time budget
    (at least 2 timed runs) -> (median, timed runs, warm-up runs)."""
    t0 = time.perf_counter()
    fn()
    first = time.perf_counter() - t0
    w_done = 1
    while w_done < warmup and (w_done + 1) * first < 0.25 * budget_s:
        fn()
        w_done += 1
    n = int(max(2, min(iters, (budget_s - w_done * first) / max(first, 1e-4))))
    times = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    return float(np.median(times)), n, w_done


def cpu_baseline(inputs, w, gpu_pooled=None, budget_s=24.0):
    """Reference-schedule torch-CPU forward (oracle/torch_ref.py) of the same workload on the host cores this
    process may use: batch 32 (the reference's own batch, train_viscosity.py:332) with 5
    warm-up + 20 timed iterations, and batch 4096 (the GPU workload) with as many of the 5 + 20 iterations as fit a
    time budget (the (B,E,D,D) tensor the reference materialises is 1.3 GB per layer call at this batch).
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
    med32, n32, w32 = _time_cpu(lambda: TR.pooled_pair(w, b32), 5, 20, 0.25 * budget_s)
    medB, nB, wB = _time_cpu(lambda: TR.pooled_pair(w, inputs), 5, 20, 0.75 * budget_s)
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in Path("/proc/cpuinfo").read_text().splitlines()
                         if l.startswith("model name"))
    except (OSError, StopIteration):
        cpu_model = "unknown"
    out = {"value": B / medB, "unit": "graph-pairs/s", "cores": cores, "kind": "port", "cpu_model": cpu_model,
           "sample": f"batch {B}: {wB} warm-up + {nB} timed iterations (median; BASELINE.md asks for 5 + 20, bounded here "
                     f"by a {0.75 * budget_s:.0f} s budget); torch-CPU fp32, {cores} threads, reference op schedule "
                     f"materialising (B,E,D,D); restatement, not TensorFlow itself",
           "batch32": {"value": 32 / med32, "unit": "graph-pairs/s", "ms_per_batch": med32 * 1e3,
                       "sample": f"batch 32: {w32} warm-up + {n32} timed iterations (median)"},
           f"batch{B}": {"value": B / medB, "unit": "graph-pairs/s", "ms_per_batch": medB * 1e3}}
    if acc:
        out["gpu_vs_port_fp64"] = acc
    return out


def time_other_configs(dev, Va, Vb):
    """Labelled extras of the bench line (never `value`): whole-model forward of BASELINE.json configs[2]
    (melting point, K = D^2 = 1024, S = 4, batch 8192: train_melting_point.py:146-198) and of configs[4]'s forward
    shape (atom_dim 128, 6 steps, batch 4096: the validation / predict path of train_viscosity.py with atom_dim=128),
    one stream, synthetic graphs of the bench's generator, with the executed exact-f32 flops beside the time."""
    import numpy as np
    import torch
    from ionic_mpnn_amd import model, synthetic, weights
    out = {}

    def timed(fn, iters):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    def executed_flops(inp, D, S):
        rows = edges = 0
        for pfx in ("cat", "an"):
            ids, conn = inp[f"{pfx}_atom"], inp[f"{pfx}_connectivity"]
            ok = (conn[:, :, 0] > 0) & (conn[:, :, 1] > 0)
            last_id = np.where(ids > 0, np.arange(ids.shape[1])[None, :] + 1, 0).max(axis=1)
            last_e = np.where(ok, conn.max(axis=2) + 1, 0).max(axis=1)
            rows += int(np.maximum(last_id, last_e).sum())
            edges += int(ok.sum())
        return S * (12 * D * D * rows + 2 * D * D * edges)

    for name, B, D, K, S, kind in (("config3_melting_point_D32_K1024_S4_B8192", 8192, 32, 1024, 4, "melting_point"),
                                   ("config5_forward_D128_K8_S6_B4096", 4096, 128, 8, 6, "viscosity")):
        inp = synthetic.make_batch(B, seed=0, with_temperature=(kind == "viscosity"))
        if kind == "melting_point":
            m = model.build_melting_point_model(Va, Vb, atom_dim=D, num_steps=S, device=dev)
        else:
            m = model.build_model(Va, Vb, atom_dim=D, bond_dim=K, num_steps=S, device=dev)
        m.load_weights(weights.init_weights(kind, Va, Vb, atom_dim=D, bond_dim=K, num_steps=S, seed=1))
        d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
        ms = timed(lambda: m(d), 10)
        fl = executed_flops(inp, D, S)
        out[name] = {"ms_per_forward": ms, "graph_pairs_per_s": B / (ms * 1e-3),
                     "encoder": m.resolve_encoder_mode(inp["cat_atom"].shape[1], inp["cat_bond"].shape[1]),
                     "executed_f32_tflops": fl / (ms * 1e-3) / 1e12,
                     "frac_of_f32_mfma_peak": fl / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                     "note": "whole model forward incl. plan kernels and the head, one stream; executed flops = exact-f32 "
                             "products on kept rows / valid edges (12 D^2 per row, 2 D^2 per edge and step)"}
        del m, d
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="graph pairs per GPU")
    ap.add_argument("--mp-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the labelled extras for BASELINE.json configs[2] and configs[4]'s forward shape")
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
                    help="schedule / arithmetic of the fused encoder (include/impnn.h); auto = exact f32: the "
                         "per-bond-type form f32t, else the pull form f32.  f16x2 (narrower products) on request only")
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

    N, E, D, K, S, B = 40, 80, 32, 8, args.mp_steps, args.batch
    # every rank draws its own shard of the global batch (seed offset by rank); weights replicated
    inputs = synthetic.make_batch(B, max_atoms=N, max_edges=E, seed=0 + rank)
    w = weights.init_weights("viscosity", synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, atom_dim=D, bond_dim=K,
                             num_steps=S, seed=1)
    m = model.build_model(synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, atom_dim=D, bond_dim=K, num_steps=S, device=dev)
    m.load_weights(w)
    m.encoder_mode = args.mode
    mode_used = m.resolve_encoder_mode(N, E) if args.schedule == "fused" else "layered"
    d_in = {k: torch.from_numpy(v).to(dev) for k, v in inputs.items()}  # resident in HBM before timing
    fused = args.schedule == "fused"

    pipelined = fused and args.pipeline and S > 0

    # --streams n: consecutive batches go to n HIP streams in turn, so the plan kernels and the uneven tail of one
    # batch's persistent encoder overlap the next batch's kernels; every step still plans and encodes one full batch
    enc_wgs = args.encoder_workgroups
    if enc_wgs is None:
        enc_wgs = 128 if (args.streams >= 3 and args.schedule == "fused" and not args.pipeline) else 0
    m.encoder_workgroups = enc_wgs
    state = {"plan": m.plan_batch(d_in) if pipelined else None}
    lanes = ([torch.cuda.Stream(device=dev) for _ in range(args.streams)]
             if args.streams > 1 and fused and not pipelined else [])
    for ln in lanes:
        ln.wait_stream(torch.cuda.current_stream(dev))
    counter = {"i": 0}

    def step():
        if lanes:
            ln = lanes[counter["i"] % len(lanes)]
            counter["i"] += 1
            with torch.cuda.stream(ln):
                return m.encode_pooled(d_in, fused=fused)
        if not pipelined:
            return m.encode_pooled(d_in, fused=fused)
        nxt = m.plan_batch(d_in)  # the next step's batch (same synthetic graphs, planned again from scratch)
        out = m.encode_pooled(d_in, plan=state["plan"])
        state["plan"] = nxt
        return out

    lib = _lib.load()
    # Untimed: bring the GPU to its sustained clock first.  An idle MI355X runs the first few hundred
    # launches ~13 % slower (measured: 92 us -> 80 us per encoder launch after ~30 ms of load), and the
    # default W=10 warm-up is 1 ms of work.  Same step() as the timed loop; nothing is cached across steps.
    ramp_steps = 0
    t_ramp = time.perf_counter()
    while (time.perf_counter() - t_ramp) * 1e3 < args.ramp_ms:
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
    elapsed = time.perf_counter() - t0  # this rank's K steps; the job time is the MAX over ranks (below)
    if world > 1:
        dist.barrier()

    # epilogue collective (outside the per-sample data path): global fingerprint checksum
    local_sum = torch.stack([pc.double().sum() + pa.double().sum(),
                             torch.tensor(float(B), dtype=torch.float64, device=dev)])
    if world > 1:
        t = local_sum.cpu() if rehearsal else local_sum
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
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

    overlapped_ms = None
    if fused:
        buf = (C.c_float * args.steps)()
        n = C.c_int32(0)
        _lib.check(lib.impnn_profile_collect(buf, args.steps, C.byref(n)))
        lib.impnn_profile_disable()
        if n.value:
            overlapped_ms = float(np.mean(np.frombuffer(buf, dtype=np.float32, count=n.value)))

    # the dominant kernel ALONE on the chip (one stream, one workgroup per CU): what roofline.achieved is computed from.
    # With several streams the event-bracketed duration of a launch includes the time it shares the chip with the
    # neighbouring batches' kernels, so that figure is reported as `overlapped` only.
    exclusive_ms, single_ms, full_ms, extras, other_configs = None, None, None, {}, None
    if fused and rank == 0:
        m.encoder_workgroups = 0
        for _ in range(5):
            m.encode_pooled(d_in, fused=True)
        # K back-to-back launches of the encoder kernel alone (one plan, impnn_encoder_run K times) between two HIP
        # events on the launch stream: (stop - start) / K.  An event pair around ONE launch also times the
        # dispatch / scratch set-up / end-of-kernel release on either side of it (measured: ~18 us on a 140 us
        # kernel, against rocprofv3's begin/end timestamps of the same run); per-launch pairs stay in `overlapped`.
        if S > 0:
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
            exclusive_ms = e0.elapsed_time(e1) / args.steps
            assert torch.equal(pooled[0], m.encode_pooled(d_in, fused=True)[0])
        else:
            exclusive_ms = events_ms(lambda: m.encode_pooled(d_in, fused=True), args.steps)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            m.encode_pooled(d_in, fused=True)
        torch.cuda.synchronize()
        single_ms = (time.perf_counter() - t1) / args.steps * 1e3  # plan + encoder, one batch at a time
        if world == 1:
            for _ in range(3):
                y = m(d_in, fused=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                y = m(d_in, fused=True)
            torch.cuda.synchronize()
            full_ms = (time.perf_counter() - t1) / args.steps * 1e3
            # labelled extras: the other encoder modes on the same batch, single stream, with their own error against
            # the timed mode's result (never `value`: "f16x2" is narrower arithmetic than the reference's f32)
            ref_c, ref_a = m.encode_pooled(d_in, fused=True)
            for other in ("f32t", "f32x3", "f32", "f16x2"):
                if other == mode_used:
                    continue
                m.encoder_mode = other
                if m.resolve_encoder_mode(N, E) != other:
                    continue
                oc, oa = m.encode_pooled(d_in, fused=True)
                k_ms = events_ms(lambda: m.encode_pooled(d_in, fused=True), args.steps)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    m.encode_pooled(d_in, fused=True)
                torch.cuda.synchronize()
                s_ms = (time.perf_counter() - t1) / args.steps * 1e3
                scale = float(max(ref_c.abs().max(), ref_a.abs().max()))
                extras[other] = {"kernel_ms": k_ms, "ms_per_step_single_stream": s_ms,
                                 "graph_pairs_per_s_single_stream": B / (s_ms * 1e-3),
                                 "max_rel_diff_vs_timed_mode": float(max((oc - ref_c).abs().max(),
                                                                         (oa - ref_a).abs().max())) / scale}
                if lanes:  # the same mode in the timed configuration (streams x workgroups of `value`), still an extra
                    m.encoder_workgroups = enc_wgs
                    for _ in range(3 * len(lanes)):
                        step()
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(args.steps):
                        step()
                    torch.cuda.synchronize()
                    l_ms = (time.perf_counter() - t1) / args.steps * 1e3
                    extras[other][f"graph_pairs_per_s_{len(lanes)}_streams"] = B / (l_ms * 1e-3)
                    m.encoder_workgroups = 0
            m.encoder_mode = args.mode
            if not args.no_other_configs:
                other_configs = time_other_configs(dev, synthetic.DEFAULT_VA, synthetic.DEFAULT_VB)
    m.encoder_workgroups = enc_wgs

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = total_pairs * args.steps / elapsed
    flops_launch = algorithmic_flops_per_pair(N, E, D, S) * B
    bytes_launch = algorithmic_bytes_per_pair(N, E, D) * B
    # what the kernel actually multiplies (it skips padding atoms and padding edges - exact):
    # update 12 D^2 per kept row, message 2 D^2 per valid edge (per-bond-type form) or 2 K D^2 per kept row (pull form)
    kept_rows, valid_edges = 0, 0
    for pfx in ("cat", "an"):
        ids, conn = inputs[f"{pfx}_atom"], inputs[f"{pfx}_connectivity"]
        ok = (conn[:, :, 0] > 0) & (conn[:, :, 1] > 0)
        last_id = np.where(ids > 0, np.arange(ids.shape[1])[None, :] + 1, 0).max(axis=1)
        last_e = np.where(ok, conn.max(axis=2) + 1, 0).max(axis=1)
        kept_rows += int(np.maximum(last_id, last_e).sum())
        valid_edges += int(ok.sum())
    msg_flops = 2 * D * D * valid_edges if mode_used in ("f32t", "f32x3") else 2 * K * D * D * kept_rows
    executed_flops = S * (12 * D * D * kept_rows + msg_flops)
    arith = {"f16x2": "f32 in/out/accumulate; every f32 GEMM product formed from fp16 hi/lo splits (3 "
                      "v_mfma_f32_16x16x32_f16 per f32 product, product error ~2^-21: NARROWER than the reference's f32)",
             "f32": "exact f32 products on v_mfma_f32_16x16x4_f32 (pull form: agg = sum_k W_k G_k)",
             "f32t": "exact f32 products: per-bond-type messages (models/layers.py:108-112 in the reference's own order) "
                     "on v_mfma_f32_4x4x1_16b_f32, GatedUpdate on v_mfma_f32_16x16x4_f32; f32 accumulate, f32 in/out",
             "f32x3": "as f32t, with the GatedUpdate GEMMs on the bf16 matrix pipe: every f32 operand carried exactly as three "
                      "bf16 terms, all nine cross products accumulated in f32 (9 v_mfma_f32_16x16x32_bf16 per 8 f32 MFMAs) - "
                      "the f32 products themselves, summed in another order; opt-in, not the default",
             "layered": "f32, one launch per reference layer"}[mode_used]
    out = {
        "metric": "molecule-graph pairs/sec (fwd), batch 4096 per MI355X",  # BASELINE.json's metric; 1 pair = 2 graphs
        "value": value, "unit": "graph-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f16x2": "f16x2 (split-fp16 products, f32 accumulate)",
                  "f32x3": "f32 (GatedUpdate products as exact bf16x3 triples on the bf16 pipe, f32 accumulate)"}.get(mode_used, "f32"),
        "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[1]: message-passing forward (embedding gather -> {S}x"
                               f"(BondMatrixMessage, Reduce, GatedUpdate) -> GlobalSumPool), cation+anion, synthetic "
                               f"padded graphs N<={N} E<={E}, D={D}, K={K}, batch {B} pairs/GPU, schedule={args.schedule}",
                   "arithmetic": arith,
                   "mode": mode_used,
                   "pipeline": ("plan kernels of step i+1 run on a side stream under the encoder of step i; every "
                                "step still plans and encodes one full batch") if pipelined else "none",
                   "streams": (f"{len(lanes)} HIP streams take consecutive batches in turn (each batch: plan + encoder in "
                               "stream order on its own workspace); kernels of neighbouring batches overlap; "
                               f"{enc_wgs or 'one per CU:'} persistent workgroups per encoder launch") if lanes
                   else "1 (every launch on one stream)",
                   "resident_batch": "every step plans and encodes the SAME resident batch from scratch (nothing is "
                                     "cached across steps); its 10 MB of inputs stay warm in L2/MALL - immaterial here: "
                                     "the path is compute-bound (HBM < 1 % of 8 TB/s)",
                   "global_batch": int(total_pairs), "molecule_graphs_per_s": 2.0 * value, "parallelism": f"batch-sharded x{world}, weights replicated, "
                   "no data-path collective; one all-reduce of the fingerprint checksum after the timed region",
                   "clock_ramp": f"{ramp_steps} untimed steps ({args.ramp_ms:g} ms) before the {args.warmup} warm-up steps, "
                                 "so that the timed steps run at the sustained GPU clock",
                   "checksum": float(local_sum[0].item())},
    }
    if single_ms:
        out["config"]["single_stream"] = {"ms_per_step": single_ms, "graph_pairs_per_s": B / (single_ms * 1e-3),
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
    k_ms = exclusive_ms or overlapped_ms
    if k_ms:
        # peak of the pipe the products run on: exact-f32 modes -> dense f32 MFMA (= f32 VALU) 157.3 TFLOP/s with the
        # SURVEY 8(d) algorithmic flops; f16x2 -> its executed fp16 MFMA flops (3 per f32 product) against 2.5 PFLOP/s
        if mode_used == "f16x2":
            flops_for_frac, peak, what = 3.0 * executed_flops, PEAK_F16_MFMA_TFLOPS, "executed fp16 MFMA flops (3 per f32 product)"
        elif mode_used == "f32x3":
            flops_for_frac, peak, what = (9.0 * S * 12 * D * D * kept_rows, PEAK_F16_MFMA_TFLOPS,
                                          "executed bf16 MFMA flops of the GatedUpdate GEMMs (9 per f32 product; the "
                                          "messages run beside them on the f32 pipe)")
        else:
            flops_for_frac, peak, what = float(flops_launch), PEAK_F32_MFMA_TFLOPS, "SURVEY 8(d) algorithmic f32 flops"
        ach = flops_for_frac / (k_ms * 1e-3) / 1e12
        kname = "encoder_typed_kernel" if mode_used in ("f32t", "f32x3") else "encoder_fused_kernel"
        out["roofline"] = {"bound": "mfma", "kernel": kname, "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                           "frac": ach / peak, "traffic": pmc_field("hbm_bytes_per_launch", kname),
                           "kernel_ms": k_ms,
                           "traffic_source": "profiles/pmc_r*.json: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch from separate "
                                             "rocprofv3 --pmc passes of this command (tools/pmc_profile.sh) on an earlier box "
                                             "run of the same kernels - PMC collection cannot share a run with the timing",
                           "timing": (f"two HIP events around {args.steps} back-to-back launches of the kernel alone on the chip "
                                      "(one plan, impnn_encoder_run K times, one stream, one workgroup per CU; untimed "
                                      "extra launches of this run), divided by K") if exclusive_ms else
                                     "HIP events per launch inside the timed loop",
                           "note": f"achieved = {what} per launch / kernel_ms; the algorithmic count (2 S (2 D^2 E + 12 D^2 N) "
                                   "per pair) includes padding atoms and padding edge slots, which the kernel skips exactly: "
                                   "`executed_flops_per_launch` is what it multiplies",
                           "algorithmic_flops_per_launch": flops_launch,
                           "executed_flops_per_launch": executed_flops,
                           "executed_achieved": executed_flops / (k_ms * 1e-3) / 1e12,
                           "matrix_pipe_busy_frac_pmc": pmc_field("mfma_pipe_busy_frac", kname),
                           "overlapped": (None if not (lanes and overlapped_ms) else {
                               "streams": len(lanes), "kernel_ms": overlapped_ms,
                               "note": "per launch as HIP events / rocprofv3 see it inside the timed loop, while the "
                                       "neighbouring batches' kernels share the chip - not a roofline figure"}),
                           "whole_step": {"achieved": flops_launch / (ms_per_step * 1e-3) / 1e12,
                                          "frac": flops_launch / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                                          "note": "algorithmic flops / ms_per_step (plan kernels included)"},
                           "hbm": {"algorithmic_bytes_per_launch": bytes_launch,
                                   "achieved_GBs": bytes_launch / (k_ms * 1e-3) / 1e9,
                                   "frac_of_8TBs": bytes_launch / (k_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(inputs, w, gpu_pooled=(pc, pa))
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
