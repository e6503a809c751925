#!/usr/bin/env python3
"""Whole-model forward at atom_dim 128, 6 steps (BASELINE configs[4]'s forward shape) in both wide encoder modes:
exact f32 MFMA (f32t) and the GatedUpdate GEMMs as bf16x9 emulation (f32x3).  python tools/wide_mode_bench.py [--batch 4096]"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ionic_mpnn_amd import model, synthetic, weights  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
inp = synthetic.make_batch(a.batch, seed=0)
m = model.build_model(Va, Vb, atom_dim=a.dim, bond_dim=8, num_steps=a.steps, device=dev)
m.load_weights(weights.init_weights("viscosity", Va, Vb, atom_dim=a.dim, bond_dim=8, num_steps=a.steps, seed=1))
d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
out, res = {}, {}
for mode in ("f32t", "f32x3"):
    m.encoder_mode = mode
    for _ in range(3):
        y = m(d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        y = m(d)
    torch.cuda.synchronize()
    out[mode] = (time.perf_counter() - t0) / a.iters * 1e3
    res[mode] = y.double().cpu().numpy()
print(json.dumps({"batch": a.batch, "atom_dim": a.dim, "mp_steps": a.steps, "ms_per_forward": out,
                  "max_rel_diff_between_modes": float(np.abs(res["f32t"] - res["f32x3"]).max() / np.abs(res["f32t"]).max())}))
