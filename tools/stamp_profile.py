#!/usr/bin/env python3
"""Diagnostics (not a benchmark): per-workgroup s_memtime stamps of encoder_fused_kernel at the
bench configuration -> where a chunk's cycles go (prologue / each step / pool) and how evenly the
chunks fill the CUs.  Runs on the GPU box:  python tools/stamp_profile.py [--batch 4096]"""
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
ap.add_argument("--mp-steps", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
B, S = args.batch, args.mp_steps
inp = synthetic.make_batch(B, seed=0)
w = weights.init_weights("viscosity", synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, num_steps=S, seed=1)
m = model.build_model(synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, num_steps=S, device=dev)
m.load_weights(w)
d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
for _ in range(3):
    m.encode_pooled(d, fused=True)
torch.cuda.synchronize()
nwg = 2 * (B * 40 // 217 + 1)
buf = torch.zeros(nwg * 32, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.impnn_debug_set_stamp_buffer(buf.data_ptr(), buf.numel() * 8)
m.encode_pooled(d, fused=True)
torch.cuda.synchronize()
lib.impnn_debug_set_stamp_buffer(None, 0)
st = buf.cpu().numpy().reshape(nwg, 32).astype(np.uint64)
live = st[:, 7] != 0
st = st[live]
R = (st[:, 6] >> np.uint64(32)).astype(np.int64)
M = (st[:, 6] & np.uint64(0xffffffff)).astype(np.int64)
t = st.astype(np.int64)
t0 = t[:, 0].min()
pro = t[:, 1] - t[:, 0]
steps = [t[:, 2 + s] - (t[:, 1] if s == 0 else t[:, 1 + s]) for s in range(min(S, 5))]
epi = t[:, 7] - t[:, 1 + min(S, 5)]
tot = t[:, 7] - t[:, 0]
print(f"live workgroups {live.sum()} of {nwg}; rows/chunk mean {R.mean():.1f} min {R.min()} max {R.max()}; "
      f"molecules/chunk mean {M.mean():.1f}; tiles/chunk mean {np.ceil(R / 16).mean():.2f}")
print(f"cycles (s_memtime ticks) per chunk: prologue {pro.mean():.0f}  " +
      "  ".join(f"step{s} {x.mean():.0f}" for s, x in enumerate(steps)) + f"  pool {epi.mean():.0f}  total {tot.mean():.0f}")
span = t[:, 7].max() - t0
print(f"kernel span {span} ticks; sum of chunk totals / 256 CUs = {tot.sum() / 256:.0f} ticks "
      f"({tot.sum() / 256 / span:.2%} of span = CU occupancy by live chunks)")
tiles = np.ceil(R / 16)
ideal = tiles * 224 * 32 / 4  # MFMA-bound cycles per step for the chunk (4 SIMDs)
print(f"ideal MFMA cycles per step per chunk {ideal.mean():.0f} vs measured step mean {np.mean([x.mean() for x in steps]):.0f}")
pp = [t[:, 0], t[:, 16], t[:, 17], t[:, 18], t[:, 19], t[:, 20], t[:, 1]]
pn = ["P0 tables+first loads", "P1 degrees+ids+table issue", "P2 placement", "P3 scan", "P4 fill+h0+image", "P5 sort"]
print("prologue phases (cycles): " + "  ".join(f"{n} {(pp[i + 1] - pp[i]).mean():.0f}" for i, n in enumerate(pn)))
ph = t[:, 8:16]
names = ["h load+deg", "gather", "msg mfma", "gates mfma+sigmoid", "cand mfma", "tanh+LN+store"]
print("wave0 tile0 phases (cycles): " + "  ".join(f"{n} {(ph[:, i + 1] - ph[:, i]).mean():.0f}" for i, n in enumerate(names))
      + f"  tile total {(ph[:, 5] - ph[:, 0]).mean():.0f}")
print(f"wave0 done -> barrier released: {(ph[:, 7] - ph[:, 6]).mean():.0f} cycles (wave 0 waits for the slowest wave)")
ends = np.sort(t[:, 7] - t0)
starts = np.sort(t[:, 0] - t0)
print("start percentiles", np.percentile(starts, [0, 25, 50, 75, 100]).astype(int))
print("end percentiles  ", np.percentile(ends, [0, 25, 50, 75, 100]).astype(int))
