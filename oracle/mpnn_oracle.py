"""numpy restatement of the reference's message-passing forward path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  **Parity unpinned** for the
floating-point layer arithmetic: the reference ships no fixtures and its
arithmetic lives in TensorFlow/Keras, which is absent here.  Every function
below follows the cited reference lines op for op, *including* the
materialised ``(B,E,D,D)`` tensor and the sequential scatter, so that it is a
restatement and not a re-derivation.  All citations are relative to
``/root/reference``.

Every function takes ``dtype`` (``np.float64`` master, ``np.float32`` to mimic
the reference's own precision).  Integer tensors are int32 like the reference's
Keras ``Input(dtype=tf.int32)`` (train_viscosity.py:150-157).

Keras defaults that the reference relies on but does not spell out:
  * ``Dense``: ``use_bias=True``, ``y = x @ kernel + bias`` (kernel ``(in,out)``).
  * ``LayerNormalization()``: ``axis=-1``, ``epsilon=1e-3``, biased variance,
    ``gamma``/``beta`` of shape ``(D,)``.
  * ``Embedding(mask_zero=False)``: plain row lookup, row 0 is a learned row.
  * ``Dropout(0.0)`` is the identity (models/layers.py:140,156).
  * ``tf.nn.softplus(x) = log(1 + exp(x))``.
"""
from __future__ import annotations

import numpy as np

LN_EPS = 1e-3  # keras.layers.LayerNormalization default epsilon


# --------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _softplus(x):
    # log(1+exp(x)), stable
    return np.logaddexp(x, 0.0).astype(x.dtype)


def dense(x, kernel, bias, activation=None):
    """keras.layers.Dense: x @ kernel + bias (train_viscosity.py:189,197-198,204)."""
    y = x @ kernel + bias
    if activation == "relu":
        y = np.maximum(y, 0)
    elif activation is not None:
        raise ValueError(activation)
    return y


def embedding(ids, table):
    """keras Embedding(mask_zero=False) (train_viscosity.py:163-164,171-172).

    Out-of-range ids raise, as tf.gather does on CPU.
    """
    ids = np.asarray(ids)
    if ids.size and (ids.min() < 0 or ids.max() >= table.shape[0]):
        raise ValueError("embedding id out of range")
    return table[ids]


def layer_norm(x, gamma, beta, eps=LN_EPS):
    """keras LayerNormalization(axis=-1) with biased variance (models/layers.py:139,154)."""
    mean = x.mean(axis=-1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True)
    inv = 1.0 / np.sqrt(var + x.dtype.type(eps))
    return (x - mean) * inv * gamma + beta


# --------------------------------------------------------------------------
# the four custom layers (models/layers.py)
# --------------------------------------------------------------------------
def bond_matrix_message(atom_state, bond_state, connectivity, bond_transform):
    """BondMatrixMessage.call, models/layers.py:100-117.

    atom_state (B,N,D) f, bond_state (B,E,K) f, connectivity (B,E,2) int [src,tgt],
    bond_transform (K,D,D).  Returns messages (B,E,D).
    """
    B, N, D = atom_state.shape
    src_idx = connectivity[:, :, 0]  # :103
    tgt_idx = connectivity[:, :, 1]  # :104
    if src_idx.size and (src_idx.min() < 0 or src_idx.max() >= N):
        raise ValueError("src index out of range")  # tf.gather raises on CPU
    # :106  tf.gather(atom_state, src_idx, batch_dims=1)
    src_atoms = np.take_along_axis(atom_state, src_idx[:, :, None].astype(np.int64), axis=1)
    # :108  tf.tensordot(bond_state, W, axes=[[2],[0]]) -> (B,E,D,D), materialised
    bond_mats = np.tensordot(bond_state, bond_transform, axes=[[2], [0]])
    # :110-112  matmul(bond_mats, src[...,None]) -> (B,E,D)
    messages = np.matmul(bond_mats, src_atoms[..., None])[..., 0]
    # :114-115
    valid = np.logical_and(src_idx > 0, tgt_idx > 0)
    messages = messages * valid[..., None].astype(messages.dtype)
    return messages


def reduce_messages(messages, tgt_idx, atom_ref):
    """Reduce.call, models/layers.py:57-83: scatter_nd of the tgt>0 rows.

    tf.scatter_nd accumulates duplicates; on CPU it walks the updates in order,
    which np.add.at reproduces.
    """
    B, E, D = messages.shape
    N = atom_ref.shape[1]
    batch_idx = np.repeat(np.arange(B, dtype=np.int64)[:, None], E, axis=1)  # :65-66
    tgt_flat = tgt_idx.reshape(-1).astype(np.int64)  # :72
    valid = tgt_flat > 0  # :74
    b_sel = batch_idx.reshape(-1)[valid]
    t_sel = tgt_flat[valid]
    if t_sel.size and t_sel.max() >= N:
        raise ValueError("tgt index out of range")  # scatter_nd raises on CPU
    out = np.zeros((B, N, D), dtype=messages.dtype)  # :78-82
    np.add.at(out, (b_sel, t_sel), messages.reshape(-1, D)[valid])
    return out


def gated_update(atom_state, agg, p):
    """GatedUpdate.call, models/layers.py:142-156.  p: dict with Wz,bz,Wr,br,Wh,bh,gamma,beta."""
    concat = np.concatenate([atom_state, agg], axis=-1)  # :144
    z = _sigmoid(dense(concat, p["Wz"], p["bz"]))  # :146
    r = _sigmoid(dense(concat, p["Wr"], p["br"]))  # :147
    r_state = r * atom_state  # :149
    h_input = np.concatenate([r_state, agg], axis=-1)  # :150
    h_tilde = np.tanh(dense(h_input, p["Wh"], p["bh"]))  # :151
    new_state = (1 - z) * atom_state + z * h_tilde  # :153
    new_state = layer_norm(new_state, p["gamma"], p["beta"])  # :154
    new_state = new_state + atom_state  # :155
    return new_state  # :156 Dropout(0.0) == identity


def global_sum_pool(atom_features, atom_ids):
    """GlobalSumPool.call, models/layers.py:161-164."""
    mask = (atom_ids > 0).astype(atom_features.dtype)[..., None]
    return (atom_features * mask).sum(axis=1)


def bond_matrix_message_fused(atom_state, bond_state, connectivity, bond_transform_flat):
    """Signature of the orphan models/bond_matrix_message.py:21-65 (flat (K,D*D) weight,
    output aggregated (B,N,D)).  The orphan's arithmetic is dead code;
    this follows the live a4+a5 arithmetic (message mask included)."""
    B, N, D = atom_state.shape
    K = bond_state.shape[-1]
    W = bond_transform_flat.reshape(K, D, D)
    m = bond_matrix_message(atom_state, bond_state, connectivity, W)
    return reduce_messages(m, connectivity[:, :, 1], atom_state)


# --------------------------------------------------------------------------
# parameter helpers
# --------------------------------------------------------------------------
def step_params(w, prefix, i, dtype):
    g = f"{prefix}_gu_{i}"
    c = lambda a: np.asarray(a, dtype=dtype)
    return {
        "bond_transform": c(w[f"{prefix}_bmm_{i}/bond_transform"]),
        "Wz": c(w[f"{g}/dense_z/kernel"]), "bz": c(w[f"{g}/dense_z/bias"]),
        "Wr": c(w[f"{g}/dense_r/kernel"]), "br": c(w[f"{g}/dense_r/bias"]),
        "Wh": c(w[f"{g}/dense_h/kernel"]), "bh": c(w[f"{g}/dense_h/bias"]),
        "gamma": c(w[f"{g}/layernorm/gamma"]), "beta": c(w[f"{g}/layernorm/beta"]),
    }


def num_steps_of(w, prefix="cat"):
    i = 0
    while f"{prefix}_bmm_{i}/bond_transform" in w:
        i += 1
    return i


# --------------------------------------------------------------------------
# encode() and the two model heads
# --------------------------------------------------------------------------
def encode(w, prefix, atom_ids, bond_ids, conn, dtype=np.float64, trace=None, pooled_only=False):
    """encode(), train_viscosity.py:166-190 / train_melting_point.py:152-174.

    Returns the fingerprint after Dense(fp_size, relu).  If ``trace`` is a dict,
    per-layer tensors are stored in it.  ``pooled_only`` returns the GlobalSumPool
    output (before the fp Dense).
    """
    c = lambda a: np.asarray(a, dtype=dtype)
    h = embedding(atom_ids, c(w["atom_embedding"]))  # :171
    bond_emb = embedding(bond_ids, c(w["bond_embedding"]))  # :172
    if trace is not None:
        trace[f"{prefix}/h0"] = h
        trace[f"{prefix}/bond_emb"] = bond_emb
    for i in range(num_steps_of(w, prefix)):  # :176
        p = step_params(w, prefix, i, dtype)
        m = bond_matrix_message(h, bond_emb, conn, p["bond_transform"])  # :178
        agg = reduce_messages(m, conn[:, :, 1], h)  # :182
        h = gated_update(h, agg, p)  # :184
        if trace is not None:
            trace[f"{prefix}/m{i}"] = m
            trace[f"{prefix}/agg{i}"] = agg
            trace[f"{prefix}/h{i + 1}"] = h
    pooled = global_sum_pool(h, atom_ids)  # :187
    if trace is not None:
        trace[f"{prefix}/pooled"] = pooled
    if pooled_only:
        return pooled
    fp = dense(pooled, c(w[f"{prefix}_fp/kernel"]), c(w[f"{prefix}_fp/bias"]), "relu")  # :189
    if trace is not None:
        trace[f"{prefix}/fp"] = fp
    return fp


def _mixed(w, inputs, dtype, trace):
    c = lambda a: np.asarray(a, dtype=dtype)
    fp_cat = encode(w, "cat", inputs["cat_atom"], inputs["cat_bond"], inputs["cat_connectivity"], dtype, trace)
    fp_an = encode(w, "an", inputs["an_atom"], inputs["an_bond"], inputs["an_connectivity"], dtype, trace)
    cat_proj = dense(fp_cat, c(w["cat_proj/kernel"]), c(w["cat_proj/bias"]), "relu")  # :197
    an_proj = dense(fp_an, c(w["an_proj/kernel"]), c(w["an_proj/bias"]), "relu")  # :198
    mixed = cat_proj + an_proj  # :201 AddTwoTensors (models/layers.py:47-49)
    if trace is not None:
        trace["mixed"] = mixed
    return mixed


def viscosity_forward(w, inputs, dtype=np.float64, trace=None):
    """build_model graph, train_viscosity.py:150-224; heads models/layers.py:10-49."""
    c = lambda a: np.asarray(a, dtype=dtype)
    mixed = _mixed(w, inputs, dtype, trace)
    vp = dense(mixed, c(w["visc_params/kernel"]), c(w["visc_params/bias"]))  # :204
    A = vp[:, 0:1]  # SliceParamA, layers.py:25-26
    Bp = np.clip(_softplus(vp[:, 1:2]), 0.0, 20.0)  # SliceParamB, :31-34
    Cp = np.clip(_softplus(vp[:, 2:3]), 0.1, 50.0)  # SliceParamC, :39-42
    T = c(inputs["temperature"]) / dtype(100.0)  # ScaleTemperature, :19-20
    log_eta = A + Bp / (T + Cp + dtype(1e-6))  # ComputeLogEta, :12-14
    if trace is not None:
        trace["visc_params"] = vp
        trace["log_eta"] = log_eta
    return log_eta


def melting_point_forward(w, inputs, dtype=np.float64, trace=None):
    """build_model graph, train_melting_point.py:146-208."""
    c = lambda a: np.asarray(a, dtype=dtype)
    mixed = _mixed(w, inputs, dtype, trace)  # :191-194
    x = dense(mixed, c(w["mp_hidden/kernel"]), c(w["mp_hidden/bias"]), "relu")  # :197
    out = dense(x, c(w["mp_out/kernel"]), c(w["mp_out/bias"]))  # :198
    if trace is not None:
        trace["mp_out"] = out
    return out


# --------------------------------------------------------------------------
# host-side input plumbing (pinned against utils/mp_utils.py)
# --------------------------------------------------------------------------
def pad_sequences_1d(seq_list, max_len, pad_val=0):
    """utils/mp_utils.py:12-16 / train_viscosity.py:52-59."""
    return np.array([list(s) + [pad_val] * (max_len - len(s)) for s in seq_list], dtype=np.int32)


def preprocess_edges_and_bonds(edge_list, bond_list, max_edges):
    """utils/mp_utils.py:18-45 / train_viscosity.py:76-110: add a reverse edge for every
    entry, pad with [0,0]/0 or truncate to 2*max_edges."""
    pe, pb = [], []
    for edges, bonds in zip(edge_list, bond_list):
        e2, b2 = [], []
        for (s, t), bid in zip(edges, bonds):
            e2 += [[s, t], [t, s]]
            b2 += [bid, bid]
        pe.append(e2)
        pb.append(b2)
    L = max_edges * 2
    pe = [e + [[0, 0]] * (L - len(e)) if len(e) < L else e[:L] for e in pe]
    pb = [b + [0] * (L - len(b)) if len(b) < L else b[:L] for b in pb]
    return np.array(pe, dtype=np.int32).reshape(len(pe), L, 2), np.array(pb, dtype=np.int32).reshape(len(pb), L)


def r2_numpy(y_true, y_pred):
    """utils/mp_utils.py:7-10."""
    ss_res = np.sum((y_true - y_pred) ** 2)
    ss_tot = np.sum((y_true - np.mean(y_true)) ** 2)
    return 1.0 - ss_res / (ss_tot + 1e-6)
