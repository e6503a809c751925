#!/usr/bin/env python3
"""Interleaved A/B timing of two libimpnn.so builds in ONE process on ONE device (guide rule 24):
    python tools/ab_bench.py ionic_mpnn_amd/csrc/ab/libA.so ionic_mpnn_amd/csrc/ab/libB.so [--rounds 12]
Times the full impnn_encoder_fused call (plan kernels + encoder) and the encoder kernel alone
(the library's event-pair profiler) on the bench workload; prints median / min per variant."""
import argparse
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

from ionic_mpnn_amd import _lib, model, ops, synthetic, weights

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--mode", default="f16x2")
args = ap.parse_args()

dev = torch.device("cuda:0")
B, S = args.batch, 3
inp = synthetic.make_batch(B, seed=0)
w = weights.init_weights("viscosity", synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, num_steps=S, seed=1)
m = model.build_model(synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, num_steps=S, device=dev)
m.load_weights(w)
d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
packed = m._packed_weights()
ions = [(d["cat_atom"], d["cat_bond"], d["cat_connectivity"]), (d["an_atom"], d["an_bond"], d["an_connectivity"])]
atab, btab = m.atom_emb.embeddings, m.bond_emb.embeddings
pooled = [torch.empty(B, 32, device=dev) for _ in range(2)]
ws = torch.empty(64 << 20, dtype=torch.uint8, device=dev)
arr = C.c_void_p * 2
mk = lambda ts: arr(*[t.data_ptr() for t in ts])

libs = []
for path in args.libs:
    lib = C.CDLL(str(Path(path).resolve()))
    for name, (res, argt) in _lib.SIGNATURES.items():
        if hasattr(lib, name):
            getattr(lib, name).restype = res
            getattr(lib, name).argtypes = argt
    libs.append(lib)


prep = {}


def call(lib):
    if hasattr(lib, "impnn_encoder_fused_prepared"):
        if id(lib) not in prep:
            nb = int(lib.impnn_encoder_prepared_bytes(32, S, btab.shape[0], ops.ENCODER_MODES[args.mode]))
            bufs = [torch.empty(nb, dtype=torch.uint8, device=dev) for _ in range(2)]
            for bf, pk in zip(bufs, packed):
                assert lib.impnn_encoder_prepare_weights(pk.data_ptr(), btab.data_ptr(), 32, 8, S, btab.shape[0],
                                                         ops.ENCODER_MODES[args.mode], bf.data_ptr(), nb,
                                                         torch.cuda.current_stream().cuda_stream) == 0
            prep[id(lib)] = bufs
        rc = lib.impnn_encoder_fused_prepared(2, mk([i[0] for i in ions]), mk([i[1] for i in ions]),
                                              mk([i[2] for i in ions]), atab.data_ptr(), atab.shape[0], btab.data_ptr(),
                                              btab.shape[0], mk(prep[id(lib)]), ops.ENCODER_MODES[args.mode], mk(pooled),
                                              B, 40, 80, 32, 8, S, 1e-3, 0, ws.data_ptr(), ws.numel(),
                                              torch.cuda.current_stream().cuda_stream)
        assert rc == 0, lib.impnn_last_error_string()
        return
    rc = lib.impnn_encoder_fused(2, mk([i[0] for i in ions]), mk([i[1] for i in ions]), mk([i[2] for i in ions]),
                                 atab.data_ptr(), atab.shape[0], btab.data_ptr(), btab.shape[0], mk(packed),
                                 ops.ENCODER_MODES[args.mode], mk(pooled), B, 40, 80, 32, 8, S, 1e-3, 0, ws.data_ptr(),
                                 ws.numel(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.impnn_last_error_string()


outs = []
for lib in libs:
    call(lib)
    torch.cuda.synchronize()
    outs.append([p.clone() for p in pooled])
for i in range(1, len(libs)):
    print(f"variant {i} vs 0: max abs diff {max(float((a - b).abs().max()) for a, b in zip(outs[0], outs[i])):.3e}")

tot = [[] for _ in libs]
ker = [[] for _ in libs]
for r in range(args.rounds):
    for i, lib in enumerate(libs):
        lib.impnn_profile_enable(args.iters)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            call(lib)
        e1.record()
        torch.cuda.synchronize()
        buf = (C.c_float * args.iters)()
        n = C.c_int32(0)
        lib.impnn_profile_collect(buf, args.iters, C.byref(n))
        lib.impnn_profile_disable()
        tot[i].append(e0.elapsed_time(e1) / args.iters * 1e3)
        ker[i].append(float(np.mean(np.frombuffer(buf, dtype=np.float32, count=n.value))) * 1e3)
for i, path in enumerate(args.libs):
    print(f"{Path(path).name:24s} call: median {np.median(tot[i]):7.1f} us  min {np.min(tot[i]):7.1f} us | "
          f"encoder kernel: median {np.median(ker[i]):7.1f} us  min {np.min(ker[i]):7.1f} us")
