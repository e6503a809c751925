"""Explicit-hydrogen shape (N = 160, E = 640) at atom_dim 128, 6 steps: the wide encoder's two modes and the
layer-at-a-time path; prints one JSON line (ms per encode of both ions)."""
import sys, json, torch
sys.path.insert(0, "/root/repo")
import bench
from ionic_mpnn_amd import model, synthetic, weights
dev = torch.device("cuda:0")
Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
B, N, E, D, S = 1024, 160, 640, 128, 6
inp = synthetic.make_explicit_h_batch(B, max_atoms=N, max_edges=E, seed=0)
m = model.build_model(Va, Vb, atom_dim=D, bond_dim=8, num_steps=S, device=dev)
m.load_weights(weights.init_weights("viscosity", Va, Vb, atom_dim=D, bond_dim=8, num_steps=S, seed=1))
d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
res = {}
for mode in ("f32t", "f32x3"):
    m.encoder_mode = mode
    res["wide_" + mode] = round(bench._gpu_timed(lambda: m.encode_pooled(d), 6), 4)
res["layered"] = round(bench._gpu_timed(lambda: m.encode_pooled(d, fused=False), 4), 4)
print(json.dumps(res))
