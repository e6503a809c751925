"""Config-5-shaped forward (atom_dim 128, 6 steps) through the wide encoder only: a workload for rocprofv3
(`rocprofv3 --kernel-trace --stats -- python3 tools/wide_probe.py`) and for IMPNN_LIB=<diagnostics build> A/B runs.
python tools/wide_probe.py [--batch 4096] [--dim 128] [--steps 6] [--iters 8]"""
import argparse
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ionic_mpnn_amd import model, synthetic, weights  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4096)
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--steps", type=int, default=6)
ap.add_argument("--iters", type=int, default=8)
ap.add_argument("--mode", default="auto", help="auto (exact f32) | f32x3 (GEMMs as bf16x9 emulation)")
args = ap.parse_args()
dev = torch.device("cuda:0")
Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
inp = synthetic.make_batch(args.batch, seed=0)
m = model.build_model(Va, Vb, atom_dim=args.dim, bond_dim=8, num_steps=args.steps, device=dev)
m.load_weights(weights.init_weights("viscosity", Va, Vb, atom_dim=args.dim, bond_dim=8, num_steps=args.steps, seed=1))
d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
m.encoder_mode = args.mode
for _ in range(3):
    m.encode_pooled(d, fused=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.iters):
    pc, pa = m.encode_pooled(d, fused=True)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / args.iters * 1e3
print(json.dumps({"mode": args.mode, "encode_ms": ms, "pairs_per_s": args.batch / ms * 1e3, "checksum": float(pc.sum() + pa.sum())}))
