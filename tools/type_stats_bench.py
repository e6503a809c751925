import sys, json, torch, numpy as np
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
    res[name] = round(bench._gpu_timed(lambda: m.encode_pooled(d), 20), 4)
# config 2 shape with six types as well
inp = synthetic.make_batch(4096, seed=0)
rng = np.random.default_rng(3)
for p in ("cat", "an"):
    b = inp[p + "_bond"]
    draw = 1 + rng.choice(6, size=b.shape, p=[0.55, 0.2, 0.12, 0.07, 0.04, 0.02])
    # both directions of a bond carry the same id (adjacent slots)
    draw[:, 1::2] = draw[:, 0::2]
    inp[p + "_bond"] = np.where(b > 0, draw, 0).astype(np.int32)
m = model.build_model(Va, Vb, num_steps=3, device=dev)
m.load_weights(weights.init_weights("viscosity", Va, Vb, num_steps=3, seed=1))
d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
for mode in ("f32t", "f32x3"):
    m.encoder_mode = mode
    res["config2_six_types_" + mode] = round(bench._gpu_timed(lambda: m.encode_pooled(d), 50), 4)
print(json.dumps(res))

# the training step (forward + backward + Adam) of config 2's model under both type statistics
from ionic_mpnn_amd import train
res2 = {}
for name, six in (("uniform71", False), ("six_types", True)):
    inp = synthetic.make_batch(4096, seed=0)
    if six:
        rng = np.random.default_rng(3)
        for p in ("cat", "an"):
            b = inp[p + "_bond"]
            draw = 1 + rng.choice(6, size=b.shape, p=[0.55, 0.2, 0.12, 0.07, 0.04, 0.02])
            draw[:, 1::2] = draw[:, 0::2]
            inp[p + "_bond"] = np.where(b > 0, draw, 0).astype(np.int32)
    y = np.random.default_rng(0).normal(4.0, 1.0, size=4096).astype(np.float32)
    m = model.build_model(Va, Vb, num_steps=3, device=dev)
    m.load_weights(weights.init_weights("viscosity", Va, Vb, num_steps=3, seed=1))
    m.compile(train.Adam(1e-3, clipnorm=1.0))
    d = m._to_device(inp)
    res2["train_step_ms_" + name] = round(bench._gpu_timed(lambda: m.train_on_batch(d, y), 10, warm=3), 4)
print(json.dumps(res2))
