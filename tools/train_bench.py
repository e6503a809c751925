"""Training step timing (not the headline metric): forward + backward + Adam(clipnorm) on
synthetic padded graphs.  python tools/train_bench.py [--batch 32] [--atom-dim 32] [--steps 3] [--iters 30]"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ionic_mpnn_amd import model, synthetic, train, weights  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--atom-dim", type=int, default=32)
ap.add_argument("--bond-dim", type=int, default=8)
ap.add_argument("--steps", type=int, default=3, help="message-passing steps")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--graph", action="store_true", help="replay one captured hipGraph per step (train.GraphedTrainStep)")
ap.add_argument("--explicit-h", action="store_true",
                help="the padded shape of the reference's real data sets (N = 160, E = 640) instead of N = 40, E = 80")
a = ap.parse_args()
dev = torch.device("cuda:0")
D, K, S, B = a.atom_dim, a.bond_dim, a.steps, a.batch
inp = synthetic.make_explicit_h_batch(B, seed=0) if a.explicit_h else synthetic.make_batch(B, seed=0)
y = np.random.default_rng(0).normal(4.0, 1.0, size=B).astype(np.float32)
m = model.build_model(synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, atom_dim=D, bond_dim=K, num_steps=S, device=dev)
m.load_weights(weights.init_weights("viscosity", synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, atom_dim=D, bond_dim=K,
                                    num_steps=S, seed=1))
m.compile(train.Adam(1e-3, clipnorm=1.0))
d = m._to_device(inp)
losses = []
step = train.GraphedTrainStep(m, d, y) if a.graph else m.train_on_batch
for _ in range(5):
    losses.append(float(step(d, y)))
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    loss = step(d, y)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / a.iters * 1e3
losses.append(float(loss))
print(json.dumps({"graph": bool(a.graph), "batch": B, "shape": "N160_E640" if a.explicit_h else "N40_E80", "atom_dim": D, "bond_dim": K, "mp_steps": S, "ms_per_train_step": ms,
                  "pairs_per_s": B / (ms * 1e-3), "loss_first": losses[0], "loss_last": losses[-1],
                  "params": int(sum(t.numel() for _, t in m.trainable_variables()))}))
