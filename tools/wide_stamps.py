#!/usr/bin/env python3
"""Diagnostics (not a benchmark): s_memtime stamps of the wide encoder's GatedUpdate and message kernels at the
config-5 shape, from a build with -DIMPNN_DIAG_WIDE_STAMPS (IMPNN_LIB=<that build>): cycles per phase of a 64-row
tile, the shader clock the kernel ran at, tiles per CU.   IMPNN_LIB=... python tools/wide_stamps.py"""
import argparse
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch

from ionic_mpnn_amd import _lib, model, synthetic, weights

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--mode", default="auto")
args = ap.parse_args()
dev = torch.device("cuda:0")
Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
inp = synthetic.make_batch(args.batch, seed=0)
m = model.build_model(Va, Vb, atom_dim=args.dim, bond_dim=8, num_steps=args.steps, device=dev)
m.load_weights(weights.init_weights("viscosity", Va, Vb, atom_dim=args.dim, bond_dim=8, num_steps=args.steps, seed=1))
d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
m.encoder_mode = args.mode
for _ in range(20):
    m.encode_pooled(d, fused=True)
torch.cuda.synchronize()
gu_grid = (2 * args.batch * 40 + 2 * 128 + 127) // 128 * 128 // 64  # the row space in 64-row tiles (ws_layout)
cus = 256
buf = torch.zeros((gu_grid + cus) * 8, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.impnn_debug_set_stamp_buffer(buf.data_ptr(), buf.numel() * 8)
m.encode_pooled(d, fused=True)
torch.cuda.synchronize()
lib.impnn_debug_set_stamp_buffer(None, 0)
st = buf.cpu().numpy().astype(np.int64).reshape(-1, 8)
gu, msg = st[:gu_grid], st[gu_grid:]
live = gu[:, 4] != 0
g = gu[live]
if len(g) == 0:
    print("no stamps: the library was not built with -DIMPNN_DIAG_WIDE_STAMPS")
    sys.exit(0)
tot = g[:, 4] - g[:, 0]
clk = tot / np.maximum(g[:, 6] - g[:, 5], 1) * 100.0  # s_memrealtime ticks at 100 MHz -> MHz
med = lambda x: float(np.median(x))
print(f"GatedUpdate tiles: {len(g)}; cycles per tile median {med(tot):.0f} (min {tot.min()}, max {tot.max()}); "
      f"shader clock median {med(clk):.0f} MHz")
print(f"  prologue {med(g[:, 1] - g[:, 0]):.0f}  phase 1 {med(g[:, 2] - g[:, 1]):.0f}  gates + phase 2 "
      f"{med(g[:, 3] - g[:, 2]):.0f}  epilogue {med(g[:, 4] - g[:, 3]):.0f}")
if (g[:, 7] != 0).all():
    print(f"  epilogue: blend + LayerNorm statistics {med(g[:, 7] - g[:, 3]):.0f}, normalise + store {med(g[:, 4] - g[:, 7]):.0f}")
span = (g[:, 4].max() - g[:, 0].min())
print(f"  launch span {span} cycles = {span / med(clk):.1f} us; sum of tiles / 256 CUs = {tot.sum() / 256:.0f} cycles")
live = msg[:, 4] != 0
q = msg[live]
if len(q):
    tot = q[:, 4] - q[:, 0]
    clk = tot / np.maximum(q[:, 6] - q[:, 5], 1) * 100.0
    print(f"message workgroups: {len(q)}; cycles median {med(tot):.0f} max {tot.max()}; tiles per workgroup {med(q[:, 7]):.0f}; "
          f"per tile {med(tot / np.maximum(q[:, 7], 1)):.0f}; prologue {med(q[:, 1] - q[:, 0]):.0f}; clock {med(clk):.0f} MHz")
