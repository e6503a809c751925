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

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 dense peak
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
    return max(1, min(n, int(os.environ.get("IMPNN_BENCH_CORES", "16"))))  # GPU-box share: 16 cores per GPU


def pmc_field(name):
    """A figure for the encoder kernel from the committed rocprofv3 --pmc passes (profiles/pmc_*.json,
    latest round wins), or None.  hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024: the gfx950
    FETCH_SIZE correction of MI355X_MICROARCH.md's HBM section."""
    best = None
    for f in sorted((ROOT / "profiles").glob("pmc_*.json")):
        try:
            best = json.loads(f.read_text()).get("encoder_fused", {}).get(name, best)
        except (OSError, ValueError):
            pass
    return best


def cpu_baseline(inputs, w, gpu_pooled=None, budget_s=20.0):
    """Reference-schedule torch-CPU forward (oracle/torch_ref.py) on a bounded sample of the same
    workload on the usable host cores.  Reported beside the GPU number; never `value`.  The same sample
    also checks the GPU result of the timed configuration (BASELINE.json: "fp32 max-abs-err vs ref"): the
    fp64 run of the port is the reference, the errors are those of the first `sample` pairs."""
    from oracle import torch_ref as TR
    cores = usable_cores()
    torch.set_num_threads(cores)
    sample = 256
    sub = {k: v[:sample] for k, v in inputs.items()}
    acc = None
    if gpu_pooled is not None:
        ref = [t.numpy() for t in TR.pooled_pair(w, sub, torch.float64)]
        got = [t[:sample].double().cpu().numpy() for t in gpu_pooled]
        abs_err = max(float(np.abs(g - r).max()) for g, r in zip(got, ref))
        rel_err = max(float(np.abs(g - r).max() / np.abs(r).max()) for g, r in zip(got, ref))
        acc = {"max_abs_err": abs_err, "max_rel_err": rel_err,
               "of": f"GlobalSumPool outputs of the first {sample} pairs vs the fp64 CPU port (tolerance 1e-5 relative)"}
    t0 = time.perf_counter()
    TR.pooled_pair(w, sub)  # warm-up / page-in
    first = time.perf_counter() - t0
    iters = int(max(2, min(50, (budget_s - first) / max(first, 1e-3))))
    times = []
    for _ in range(iters):
        t0 = time.perf_counter()
        TR.pooled_pair(w, sub)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    out = {"value": sample / med, "unit": "graph-pairs/s", "cores": cores, "kind": "port",
           "sample": f"{sample} pairs x {iters} iterations (median), torch-CPU fp32, reference op schedule "
                     f"materialising (B,E,D,D); restatement, not TensorFlow itself"}
    if acc:
        out["gpu_vs_port_fp64"] = acc
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="graph pairs per GPU")
    ap.add_argument("--mp-steps", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--schedule", choices=["fused", "layered"], default="fused")
    ap.add_argument("--pipeline", action="store_true",
                    help="enqueue the plan kernels of step i+1 on a side stream before the encoder of step i "
                         "(default: plan and encode every batch back to back on one stream, which is faster on "
                         "MI355X: the encoder fills every CU's register file, see include/impnn.h)")
    ap.add_argument("--streams", type=int, default=3,
                    help="HIP streams that take consecutive batches in turn (fused schedule; 1: every launch on one "
                         "stream).  Measured on MI355X (M pairs/s): 1 stream x 256 workgroups 39.2, 2 x 256 44.2, "
                         "3 x 256 45.4, 2 x 192 47.0, 3 x 128 49.0-49.5, 3 x 160 47.7, 3 x 96 47.4, 4 x 128 39.1")
    ap.add_argument("--encoder-workgroups", type=int, default=None,
                    help="persistent workgroups per encoder launch (impnn_encoder_set_workgroups; 0 = one per CU). "
                         "Default: 128 with 3 or more streams, else 0")
    ap.add_argument("--ramp-ms", type=float, default=150.0,
                    help="untimed clock ramp before the W warm-up steps: the same step() repeated for this many "
                         "milliseconds (0 disables)")
    ap.add_argument("--mode", choices=["auto", "f32t", "f32", "f16x2"], default="auto",
                    help="GEMM arithmetic of the fused encoder (include/impnn.h); auto = f16x2 when the static "
                         "range bound holds, else exact f32")
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

    # extra (not `value`): the whole model forward = hot path + impnn_model_head, same batch
    full_ms = None
    if fused and lanes:
        m.encoder_workgroups = 0  # the single-stream extras below run alone on the chip: one workgroup per CU
    if fused and world == 1:
        for _ in range(3):
            y = m(d_in, fused=True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            y = m(d_in, fused=True)
        torch.cuda.synchronize()
        full_ms = (time.perf_counter() - t1) / args.steps * 1e3

    kernel_ms = None
    if fused:
        buf = (C.c_float * args.steps)()
        n = C.c_int32(0)
        _lib.check(lib.impnn_profile_collect(buf, args.steps, C.byref(n)))
        lib.impnn_profile_disable()
        if n.value:
            kernel_ms = float(np.mean(np.frombuffer(buf, dtype=np.float32, count=n.value)))

    # the same kernel without a neighbour: K more steps on ONE stream, outside the timed region.  With several streams
    # the event-bracketed duration of a launch includes the time it shares the chip with the other batch's kernels.
    exclusive_ms = None
    if fused and lanes and rank == 0:
        _lib.check(lib.impnn_profile_enable(args.steps))
        for _ in range(args.steps):
            m.encode_pooled(d_in, fused=True)
        torch.cuda.synchronize()
        buf = (C.c_float * args.steps)()
        n = C.c_int32(0)
        _lib.check(lib.impnn_profile_collect(buf, args.steps, C.byref(n)))
        lib.impnn_profile_disable()
        if n.value:
            exclusive_ms = float(np.mean(np.frombuffer(buf, dtype=np.float32, count=n.value)))
    m.encoder_workgroups = enc_wgs

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    value = total_pairs * args.steps / elapsed
    flops_launch = algorithmic_flops_per_pair(N, E, D, S) * B
    bytes_launch = algorithmic_bytes_per_pair(N, E, D) * B
    out = {
        "metric": "molecule-graph pairs/sec (fwd), batch 4096 per MI355X",  # BASELINE.json's metric; 1 pair = 2 graphs
        "value": value, "unit": "graph-pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[1]: message-passing forward (embedding gather -> {S}x"
                               f"(BondMatrixMessage, Reduce, GatedUpdate) -> GlobalSumPool), cation+anion, synthetic "
                               f"padded graphs N<={N} E<={E}, D={D}, K={K}, batch {B} pairs/GPU, schedule={args.schedule}",
                   "arithmetic": {"f16x2": "f32 in/out/accumulate; every f32 GEMM product formed from fp16 hi/lo splits "
                                           "(3 v_mfma_f32_16x16x32_f16 per f32 product, error ~2^-21; parity <=1e-5 vs "
                                           "fp64 oracle in tests/test_gpu_encoder.py)",
                                  "f32": "exact f32 products on v_mfma_f32_16x16x4_f32 (pull form)",
                                  "f32t": "exact f32 products: per-bond-type messages on v_mfma_f32_4x4x1_16b_f32, "
                                          "GatedUpdate on v_mfma_f32_16x16x4_f32",
                                  "layered": "f32 VALU, one launch per reference layer"}[mode_used],
                   "mode": mode_used,
                   "pipeline": ("plan kernels of step i+1 run on a side stream under the encoder of step i; every "
                                "step still plans and encodes one full batch") if pipelined else "none",
                   "streams": (f"{len(lanes)} HIP streams take consecutive batches in turn (each batch: plan + encoder in "
                               "stream order on its own workspace); kernels of neighbouring batches overlap; "
                               f"{enc_wgs or 'one per CU:'} persistent workgroups per encoder launch") if lanes
                   else "1 (every launch on one stream)",
                   "global_batch": int(total_pairs), "molecule_graphs_per_s": 2.0 * value, "parallelism": f"batch-sharded x{world}, weights replicated, "
                   "no data-path collective; one all-reduce of the fingerprint checksum after the timed region",
                   "clock_ramp": f"{ramp_steps} untimed steps ({args.ramp_ms:g} ms) before the {args.warmup} warm-up steps, "
                                 "so that the timed steps run at the sustained GPU clock",
                   "checksum": float(local_sum[0].item())},
    }
    if full_ms:
        out["config"]["full_model_forward"] = {"ms_per_step": full_ms, "graph_pairs_per_s": B / (full_ms * 1e-3),
                                               "note": "hot path + heads (impnn_model_head) -> log_eta; not `value`"}
    if rehearsal:
        out["config"]["rehearsal"] = f"{world} ranks share {ndev} GPU(s) over gloo - not a scaling number"
    if kernel_ms:
        ach = flops_launch / (kernel_ms * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": "encoder_fused_kernel", "achieved": ach,
                           "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA_TFLOPS,
                           "traffic": pmc_field("hbm_bytes_per_launch"), "kernel_ms": kernel_ms,
                           "matrix_pipe_busy_frac_pmc": pmc_field("mfma_pipe_busy_frac"),
                           "note": "achieved = SURVEY 8(d) algorithmic f32 flops / measured kernel time; peak = dense "
                                   "f32 MFMA (= f32 VALU) peak, the rate an exact-f32 implementation is bound by. "
                                   "In mode f16x2 the products run on the fp16 matrix pipe (3 per f32 product), so "
                                   "frac may exceed 1; the kernel is then VALU-issue bound (DESIGN.md 4.1)",
                           "algorithmic_flops_per_launch": flops_launch,
                           "overlap": (None if not lanes else {
                               "streams": len(lanes),
                               "note": "kernel_ms / achieved / frac above are per launch as HIP events and rocprofv3 see "
                                       "it while neighbouring batches' kernels share the chip; `exclusive` is the same "
                                       "kernel alone on one stream with one workgroup per CU (K untimed extra steps of "
                                       "this run); `whole_step` divides "
                                       "the algorithmic flops by ms_per_step (plan kernels included)",
                               "exclusive": (None if not exclusive_ms else {
                                   "kernel_ms": exclusive_ms,
                                   "achieved": flops_launch / (exclusive_ms * 1e-3) / 1e12,
                                   "frac": flops_launch / (exclusive_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}),
                               "whole_step": {"achieved": flops_launch / (ms_per_step * 1e-3) / 1e12,
                                              "frac": flops_launch / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS}}),
                           "hbm": {"algorithmic_bytes_per_launch": bytes_launch,
                                   "achieved_GBs": bytes_launch / (kernel_ms * 1e-3) / 1e9,
                                   "frac_of_8TBs": bytes_launch / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(inputs, w, gpu_pooled=(pc, pa))
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
