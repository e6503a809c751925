"""Forward latency of the fused path at small batches (serving / model.predict with Keras' batch 32).
python tools/latency_bench.py [--dim 128 --steps 6] [--layered]"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ionic_mpnn_amd import model, synthetic, weights  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, default=32)
ap.add_argument("--steps", type=int, default=3)
ap.add_argument("--layered", action="store_true", help="the layer-at-a-time kernels instead of the fused / wide encoder")
ap.add_argument("--batches", type=int, nargs="+", default=[1, 32, 256, 1024, 4096, 16384])
args = ap.parse_args()
dev = torch.device("cuda:0")
Va, Vb, S = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, args.steps
m = model.build_model(Va, Vb, atom_dim=args.dim, num_steps=S, device=dev)
m.load_weights(weights.init_weights("viscosity", Va, Vb, atom_dim=args.dim, num_steps=S, seed=1))
fused = False if args.layered else None
out = {}
for B in args.batches:
    d = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(B, seed=0).items()}
    for _ in range(300):
        y = m(d, fused=fused)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        y = m(d, fused=fused)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    out[B] = {"us_per_forward": round(us, 1), "pairs_per_s": round(B / us * 1e6)}
print(json.dumps(out))
