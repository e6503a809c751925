"""Forward timing of the other BASELINE.json configurations (parity-test cases, not the bench line):
config 3 (melting point: D=32, K=D*D=1024, S=4, B=8192) and config 5's shape (D=128, K=8, S=6, B=4096).
python tools/config_bench.py"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ionic_mpnn_amd import model, synthetic, weights  # noqa: E402

dev = torch.device("cuda:0")
Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


out = {}
B = 8192
inp = synthetic.make_batch(B, seed=0, with_temperature=False)
m = model.build_melting_point_model(Va, Vb, atom_dim=32, num_steps=4, device=dev)
m.load_weights(weights.init_weights("melting_point", Va, Vb, atom_dim=32, bond_dim=1024, num_steps=4, seed=1))
d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
ms = timeit(lambda: m(d), 10)
out["config3_melting_point_D32_K1024_S4_B8192"] = {"ms": ms, "pairs_per_s": B / (ms * 1e-3)}
B = 4096
inp = synthetic.make_batch(B, seed=0)
m = model.build_model(Va, Vb, atom_dim=128, bond_dim=8, num_steps=6, device=dev)
m.load_weights(weights.init_weights("viscosity", Va, Vb, atom_dim=128, bond_dim=8, num_steps=6, seed=1))
d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
ms = timeit(lambda: m(d), 5)
out["config5_shape_D128_K8_S6_B4096_forward"] = {"ms": ms, "pairs_per_s": B / (ms * 1e-3)}
print(json.dumps(out))
