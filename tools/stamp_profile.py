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
ap.add_argument("--mode", default="auto")
args = ap.parse_args()
dev = torch.device("cuda:0")
B, S = args.batch, args.mp_steps
inp = synthetic.make_batch(B, seed=0)
w = weights.init_weights("viscosity", synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, num_steps=S, seed=1)
m = model.build_model(synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, num_steps=S, device=dev)
m.load_weights(w)
m.encoder_mode = args.mode
mode = m.resolve_encoder_mode(40, 80)
print('encoder mode:', mode)
d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
for _ in range(3):
    m.encode_pooled(d, fused=True)
torch.cuda.synchronize()
nwg = torch.cuda.get_device_properties(0).multi_processor_count
buf = torch.zeros(nwg * 32 + 16, dtype=torch.int64, device=dev)
lib = _lib.load()
lib.impnn_debug_set_stamp_buffer(buf.data_ptr(), buf.numel() * 8)
m.encode_pooled(d, fused=True)
torch.cuda.synchronize()
lib.impnn_debug_set_stamp_buffer(None, 0)
allb = buf.cpu().numpy().astype(np.int64)
ps = allb[nwg * 32:]
idx = [i for i in range(16) if ps[i] != 0]
print('plan_chunks workgroup 0 stamps (cycles from start):', [(i, int(ps[i] - ps[0])) for i in idx])
st = allb[:nwg * 32].reshape(nwg, 32)
live = st[:, 7] != 0
st = st[live]
tot = st[:, 7] - st[:, 0]
pro, steps, pool, nch = st[:, 1], st[:, 2], st[:, 3], st[:, 4]
print(f"persistent workgroups with work: {live.sum()} of {nwg}; chunks per workgroup mean {nch.mean():.2f} min {nch.min()} max {nch.max()}")
print(f"cycles per workgroup: total mean {tot.mean():.0f} max {tot.max()} min {tot.min()} | prologues {pro.mean():.0f} "
      f"steps {steps.mean():.0f} pools {pool.mean():.0f}")
print(f"per chunk: prologue {(pro / nch).mean():.0f}  steps {(steps / nch).mean():.0f}  pool {(pool / nch).mean():.0f}")

if mode in ("f32t", "f32x3"):
    # typed encoder: stamp 14 = end of step 0, 16+w = wave w done with its message batches (step 1, first chunk),
    # 12 = mid-step barrier released, 15 = end of step 1
    t0 = st[:, 14]
    ok = (t0 > 0) & (st[:, 15] > 0)
    arr = (st[:, 16:32] - t0[:, None])[ok]
    mid = (st[:, 12] - t0)[ok]
    end = (st[:, 15] - t0)[ok]
    srt = np.sort(arr, axis=1)
    print(f"step 1 of the first chunk ({ok.sum()} workgroups): duration {end.mean():.0f}; message phase: waves done at "
          f"min {srt[:, 0].mean():.0f} / median {srt[:, 8].mean():.0f} / last {srt[:, 15].mean():.0f}, barrier released "
          f"{mid.mean():.0f}; atom phase + end barrier {(end - mid).mean():.0f}")
    if (st[:, 11] > 0).any():  # round 3: wave 1's atom phase in detail (its tile: one of the heaviest)
        a = st[ok]
        seg = [a[:, 8] - a[:, 12], a[:, 9] - a[:, 8], a[:, 10] - a[:, 9], a[:, 11] - a[:, 10], a[:, 15] - a[:, 11]]
        print("  wave 1, atom phase: Reduce %d | GEMMs + gates %d | epilogue %d | next-step fetches %d | wait at the end barrier %d"
              % tuple(int(x.mean()) for x in seg))
    sys.exit(0)

# one step (first chunk, step 1) in detail: when each wave reaches the step barrier, barrier + image copy cost
t0 = st[:, 14]                      # end of step 0 == start of step 1
arr = st[:, 16:32] - t0[:, None]    # per-wave arrival at the barrier
rel = st[:, 13] - t0                # all waves through the first barrier
end = st[:, 15] - t0                # image copied, second barrier passed
ok = (t0 > 0) & (st[:, 15] > 0)
arr, rel, end = arr[ok], rel[ok], end[ok]
srt = np.sort(arr, axis=1)
print(f"step 1 of the first chunk ({ok.sum()} workgroups): duration {end.mean():.0f}; waves reach the barrier at "
      f"min {srt[:, 0].mean():.0f} / 25% {srt[:, 4].mean():.0f} / median {srt[:, 8].mean():.0f} / 75% {srt[:, 12].mean():.0f} / "
      f"last {srt[:, 15].mean():.0f}; barrier released {rel.mean():.0f}; copy + 2nd barrier {(end - rel).mean():.0f}")
bysimd = arr.reshape(-1, 4, 4)      # [wg, slot, simd]  (wave = slot * 4 + simd)
print("  last arrival per SIMD (mean):", [int(x) for x in bysimd.max(axis=1).mean(axis=0)],
      " first:", [int(x) for x in bysimd.min(axis=1).mean(axis=0)])
