"""Explicit-hydrogen shape (N=160, E=640) timed in the typed encoder's two modes; prints one JSON line (ms per forward)."""
import sys, json, torch
sys.path.insert(0, "/root/repo")
import bench
from ionic_mpnn_amd import model, synthetic, weights
dev = torch.device("cuda:0")
Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
B, N, E, S = 4096, 160, 640, 3
res = {}
for name, probs in (("uniform71", None), ("six_types", (0.55, 0.2, 0.12, 0.07, 0.04, 0.02))):
    inp = synthetic.make_explicit_h_batch(B, max_atoms=N, max_edges=E, seed=0, bond_type_probs=probs)
    m = model.build_model(Va, Vb, atom_dim=32, bond_dim=8, num_steps=S, device=dev)
    m.load_weights(weights.init_weights("viscosity", Va, Vb, atom_dim=32, bond_dim=8, num_steps=S, seed=1))
    d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
    outs = {}
    for mode in ("f32t", "f32x3"):
        m.encoder_mode = mode
        outs[mode] = m.encode_pooled(d)
        res[name + "_" + mode] = round(bench._gpu_timed(lambda: m.encode_pooled(d), 20), 4)
    a, b = outs["f32t"], outs["f32x3"]
    a = torch.cat([t.flatten() for t in a]) if isinstance(a, (tuple, list)) else a
    b = torch.cat([t.flatten() for t in b]) if isinstance(b, (tuple, list)) else b
    res[name + "_max_rel"] = float(((a - b).abs() / (a.abs() + 1e-3)).max())
print(json.dumps(res))
