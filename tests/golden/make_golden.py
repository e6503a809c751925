"""Generates the oracle golden vectors (inputs, weights, per-layer fp64 outputs).

    python tests/golden/make_golden.py

The reference itself cannot run here (TensorFlow absent), so these vectors come from
oracle/mpnn_oracle.py, the op-for-op numpy restatement ("parity unpinned" for the layer
arithmetic - see oracle/__init__.py).  They freeze the oracle: a later change of the oracle or
of the generators that alters any number fails tests/test_oracle.py.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from ionic_mpnn_amd import synthetic, weights  # noqa: E402
from oracle import mpnn_oracle as O  # noqa: E402

HERE = Path(__file__).resolve().parent


def save_case(name, kind, inputs, w, extra_keys=()):
    trace = {}
    fwd = O.viscosity_forward if kind == "viscosity" else O.melting_point_forward
    out = fwd(w, inputs, np.float64, trace)
    blob = {f"in/{k}": v for k, v in inputs.items()}
    blob.update({f"w/{k}": v for k, v in w.items()})
    for k, v in trace.items():
        blob[f"out/{k}"] = np.asarray(v, dtype=np.float64)
    blob["out/final"] = np.asarray(out, dtype=np.float64)
    blob["meta/kind"] = np.array(kind)
    np.savez_compressed(HERE / f"{name}.npz", **blob)
    print(name, "final", np.asarray(out).ravel()[:4])


# tiny: B=4, N=6, E=12, D=8, K=4, S=2; biases / LayerNorm affine randomised
inp = synthetic.make_batch(4, max_atoms=6, max_edges=12, atom_vocab_size=11, bond_vocab_size=5, min_atoms=2, seed=11)
w = weights.init_weights("viscosity", 11, 5, atom_dim=8, bond_dim=4, fp_size=6, mixing_size=5, num_steps=2, seed=12,
                         perturb=True)
save_case("tiny_viscosity", "viscosity", inp, w)

# config-2 shaped (D=32, K=8, S=3, N=40, E=80), B=8, the bench's seeds (graphs 0, weights 1)
inp = synthetic.make_batch(8, seed=0)
w = weights.init_weights("viscosity", synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, num_steps=3, seed=1)
save_case("config2_b8", "viscosity", inp, w)

# same shape, perturbed biases/affine, 4 steps (reference default num_steps=4), other seed
inp = synthetic.make_batch(6, seed=5)
w = weights.init_weights("viscosity", synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, num_steps=4, seed=6, perturb=True)
save_case("config2_perturbed_b6", "viscosity", inp, w)

# melting-point head with bond_dim = atom_dim**2 (train_melting_point.py:146), small D
inp = synthetic.make_batch(5, max_atoms=10, max_edges=20, atom_vocab_size=15, bond_vocab_size=7, min_atoms=3, seed=21,
                           with_temperature=False)
w = weights.init_weights("melting_point", 15, 7, atom_dim=8, bond_dim=64, fp_size=8, mixing_size=6, num_steps=2,
                         seed=22, perturb=True)
save_case("tiny_melting_point", "melting_point", inp, w)

# wide atom states (train_viscosity.py with a larger atom_dim; the config-5 family): atom_dim 64, 2 steps, small graphs
inp = synthetic.make_batch(5, max_atoms=14, max_edges=28, atom_vocab_size=17, bond_vocab_size=6, min_atoms=3, seed=31)
w = weights.init_weights("viscosity", 17, 6, atom_dim=64, bond_dim=8, fp_size=16, mixing_size=10, num_steps=2, seed=32,
                         perturb=True)
save_case("wide_d64_b5", "viscosity", inp, w)
