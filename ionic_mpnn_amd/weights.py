"""Weight containers and Keras-default initialisers for the MPNN forward path.

Weights are a flat ``dict[str, np.ndarray(float32)]`` whose keys mirror the
reference's Keras layer/variable names so that a trained ``.keras`` file can be
mapped onto them later:

  atom_embedding (Va,D)                      Embedding, train_viscosity.py:163
  bond_embedding (Vb,K)                      Embedding, train_viscosity.py:164
  {p}_bmm_{i}/bond_transform (K,D,D)         models/layers.py:93-98, name at train_viscosity.py:178
  {p}_gu_{i}/dense_{z,r,h}/{kernel,bias}     models/layers.py:136-138  kernel (2D,D), bias (D,)
  {p}_gu_{i}/layernorm/{gamma,beta}          models/layers.py:139
  {p}_fp/{kernel,bias}                       Dense(fp_size, relu), train_viscosity.py:189
  {cat,an}_proj/{kernel,bias}                Dense(mixing_size, relu), train_viscosity.py:197-198
  visc_params/{kernel,bias}                  Dense(3), train_viscosity.py:204
  mp_hidden/{kernel,bias}, mp_out/{kernel,bias}   train_melting_point.py:197-198

with p in {cat, an}: weights are NOT shared between ions or steps
(train_viscosity.py:176-184), only the two embedding tables are.
"""
from __future__ import annotations

import numpy as np


def glorot_uniform(rng, shape):
    """keras.initializers.GlorotUniform: limit = sqrt(6/(fan_in+fan_out)); for rank>2 the
    leading dims count as receptive field (fan_in = shape[-2]*rf, fan_out = shape[-1]*rf)."""
    shape = tuple(int(s) for s in shape)
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    elif len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape).astype(np.float32)


def embedding_uniform(rng, shape):
    """keras Embedding default initialiser: uniform(-0.05, 0.05)."""
    return rng.uniform(-0.05, 0.05, size=shape).astype(np.float32)


def init_weights(kind, atom_vocab_size, bond_vocab_size, atom_dim=32, bond_dim=8, fp_size=32,
                 mixing_size=20, num_steps=4, seed=1, perturb=False):
    """Random-init weights of the viscosity ('viscosity') or melting-point ('melting_point') model.

    Keras defaults give zero biases and gamma=1/beta=0; ``perturb=True`` randomises those too so
    that parity tests exercise the bias / affine terms.
    """
    if kind not in ("viscosity", "melting_point"):
        raise ValueError(kind)
    rng = np.random.default_rng(seed)
    D, K = int(atom_dim), int(bond_dim)
    w = {
        "atom_embedding": embedding_uniform(rng, (atom_vocab_size, D)),
        "bond_embedding": embedding_uniform(rng, (bond_vocab_size, K)),
    }

    def bias(n):
        return (rng.uniform(-0.1, 0.1, size=n) if perturb else np.zeros(n)).astype(np.float32)

    for p in ("cat", "an"):
        for i in range(num_steps):
            w[f"{p}_bmm_{i}/bond_transform"] = glorot_uniform(rng, (K, D, D))
            for g in ("z", "r", "h"):
                w[f"{p}_gu_{i}/dense_{g}/kernel"] = glorot_uniform(rng, (2 * D, D))
                w[f"{p}_gu_{i}/dense_{g}/bias"] = bias(D)
            w[f"{p}_gu_{i}/layernorm/gamma"] = (
                rng.uniform(0.5, 1.5, size=D) if perturb else np.ones(D)).astype(np.float32)
            w[f"{p}_gu_{i}/layernorm/beta"] = bias(D)
        w[f"{p}_fp/kernel"] = glorot_uniform(rng, (D, fp_size))
        w[f"{p}_fp/bias"] = bias(fp_size)
        w[f"{p}_proj/kernel"] = glorot_uniform(rng, (fp_size, mixing_size))
        w[f"{p}_proj/bias"] = bias(mixing_size)
    if kind == "viscosity":
        w["visc_params/kernel"] = glorot_uniform(rng, (mixing_size, 3))
        w["visc_params/bias"] = bias(3)
    else:
        w["mp_hidden/kernel"] = glorot_uniform(rng, (mixing_size, fp_size))
        w["mp_hidden/bias"] = bias(fp_size)
        w["mp_out/kernel"] = glorot_uniform(rng, (fp_size, 1))
        w["mp_out/bias"] = bias(1)
    return w


def num_steps_of(w, prefix="cat"):
    i = 0
    while f"{prefix}_bmm_{i}/bond_transform" in w:
        i += 1
    return i
