"""Race hunt: the fused encoder must return bit-identical results on every one of many back-to-back launches of
the same batch (its barriers, in-place h update and LDS staging leave no room for timing-dependent results; in the
typed mode the type runs are handed out through an LDS counter, so WHICH wave computes a message varies run to run -
the message values and the order they are summed in do not).
python tools/soak_determinism.py [--iters 3000]"""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ionic_mpnn_amd import model, synthetic, weights  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=3000)
ap.add_argument("--wide", action="store_true", help="the wide encoder (atom_dim 128 / 64) instead of the D = 32 ones")
a = ap.parse_args()
dev = torch.device("cuda:0")
bad = 0
cases = ((4096, 3, "f32t", 0, 32), (4096, 3, "f32", 1, 32), (1000, 4, "f32t", 2, 32), (8192, 2, "f32t", 3, 32),
         (4096, 3, "f16x2", 4, 32))
if a.wide:
    cases = ((4096, 6, "f32t", 5, 128), (777, 3, "f32t", 6, 128), (2048, 2, "f32t", 7, 64))
for (B, S, mode, seed, D) in cases:
    m = model.build_model(synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, atom_dim=D, num_steps=S, device=dev)
    m.load_weights(weights.init_weights("viscosity", synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, atom_dim=D, num_steps=S,
                                        seed=seed, perturb=True))
    m.encoder_mode = mode
    d = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(B, seed=seed).items()}
    pc0, pa0 = [t.clone() for t in m.encode_pooled(d, fused=True)]
    mism = torch.zeros((), dtype=torch.int64, device=dev)
    for _ in range(a.iters):
        pc, pa = m.encode_pooled(d, fused=True)
        mism += (pc != pc0).any().long() + (pa != pa0).any().long()
    n = int(mism.item())
    bad += n
    print(f"B={B} S={S} D={D} mode={mode}: {a.iters} launches, {n} differing results, finite={bool(torch.isfinite(pc0).all())}")
sys.exit(1 if bad else 0)
