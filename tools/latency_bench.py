"""Forward latency of the fused path at small batches (serving / model.predict with Keras' batch 32).
python tools/latency_bench.py"""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ionic_mpnn_amd import model, synthetic, weights  # noqa: E402

dev = torch.device("cuda:0")
Va, Vb, S = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, 3
m = model.build_model(Va, Vb, num_steps=S, device=dev)
m.load_weights(weights.init_weights("viscosity", Va, Vb, num_steps=S, seed=1))
out = {}
for B in (1, 32, 256, 1024, 4096, 16384):
    d = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(B, seed=0).items()}
    for _ in range(300):
        y = m(d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 300
    for _ in range(n):
        y = m(d)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    out[B] = {"us_per_forward": round(us, 1), "pairs_per_s": round(B / us * 1e6)}
print(json.dumps(out))
