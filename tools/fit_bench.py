"""Epoch time of model.fit at the reference's batch 32 (train_viscosity.py:328-338) on a device-resident synthetic
training set: the whole loop (shuffle, batch gather, graphed step, loss bookkeeping), not only the step."""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ionic_mpnn_amd import model as MM, synthetic, train  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--samples", type=int, default=3200)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--epochs", type=int, default=6)
ap.add_argument("--atom-dim", type=int, default=32)
ap.add_argument("--steps", type=int, default=3, help="message-passing steps")
args = ap.parse_args()
Va, Vb = 124, 72
m = MM.build_model(Va, Vb, atom_dim=args.atom_dim, num_steps=args.steps, device="cuda")
m.compile(train.Adam(1e-3, clipnorm=1.0))
x = synthetic.make_batch(args.samples, max_atoms=40, max_edges=80, atom_vocab_size=Va, bond_vocab_size=Vb, seed=0)
y = np.random.default_rng(0).normal(1.0, 0.5, size=args.samples).astype(np.float32)
x = m._to_device(x)
m.fit(x, y, epochs=1, batch_size=args.batch, seed=0)  # capture + warm-up
torch.cuda.synchronize()
t0 = time.perf_counter()
h = m.fit(x, y, epochs=args.epochs, batch_size=args.batch, seed=1)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.epochs
steps = (args.samples + args.batch - 1) // args.batch
# the captured step alone (same shapes), replayed back to back: what the loop would cost with no host work at all
rows = torch.arange(args.batch, device="cuda")
g = train.GraphedTrainStep(m, {k: v[rows] for k, v in x.items()}, y[:args.batch])
for _ in range(20):
    g.graph.replay()
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(200):
    g.graph.replay()
torch.cuda.synchronize()
replay_ms = (time.perf_counter() - t1) / 200 * 1e3
print(json.dumps({"samples": args.samples, "batch": args.batch, "ms_per_epoch": dt * 1e3, "ms_per_step": dt * 1e3 / steps, "ms_per_replay_only": replay_ms,
                  "pairs_per_s": args.samples / dt, "loss": h.history["loss"]}))
