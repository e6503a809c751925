"""Second, independent CPU restatement of the reference forward: torch-CPU fp32 in the reference's
own op schedule.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py); "parity unpinned".

Two jobs:
  1. cross-check of oracle/mpnn_oracle.py (two implementations written from the same reference
     lines must agree);
  2. the ``cpu_baseline`` of bench.py: it executes what TF-CPU would - batched gather
     (models/layers.py:106), tensordot MATERIALISING (B,E,D,D) (:108), batched mat-vec (:110-112),
     mask multiply (:114-115), boolean_mask + scatter-add (:65-82), three separate Dense +
     sigmoid/tanh, LayerNorm(eps=1e-3), residual (:144-156), masked sum (:163-164).  It is a
     stand-in for "reference TF-CPU" (BASELINE.md 2), labelled kind="port".
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _t(a, dtype=torch.float32):
    return torch.as_tensor(a, dtype=dtype)


def bond_matrix_message(atom_state, bond_state, connectivity, W):
    src = connectivity[:, :, 0].long()
    tgt = connectivity[:, :, 1].long()
    D = atom_state.shape[-1]
    src_atoms = torch.gather(atom_state, 1, src[:, :, None].expand(-1, -1, D))      # :106
    bond_mats = torch.tensordot(bond_state, W, dims=([2], [0]))                      # :108 (B,E,D,D)
    messages = torch.matmul(bond_mats, src_atoms.unsqueeze(-1)).squeeze(-1)          # :110-112
    valid = (src > 0) & (tgt > 0)                                                    # :114
    return messages * valid[..., None].to(messages.dtype)                            # :115


def reduce_messages(messages, tgt_idx, num_atoms):
    B, E, D = messages.shape
    tgt = tgt_idx.long().reshape(-1)
    bidx = torch.arange(B).repeat_interleave(E)                                      # :65-66
    valid = tgt > 0                                                                  # :74
    flat = (bidx[valid] * num_atoms + tgt[valid])                                    # :75
    out = torch.zeros(B * num_atoms, D, dtype=messages.dtype)
    out.index_add_(0, flat, messages.reshape(-1, D)[valid])                          # :76-82
    return out.reshape(B, num_atoms, D)


def gated_update(h, agg, p, eps=1e-3):
    c = torch.cat([h, agg], dim=-1)                                                  # :144
    z = torch.sigmoid(c @ p["Wz"] + p["bz"])                                         # :146
    r = torch.sigmoid(c @ p["Wr"] + p["br"])                                         # :147
    hin = torch.cat([r * h, agg], dim=-1)                                            # :149-150
    ht = torch.tanh(hin @ p["Wh"] + p["bh"])                                         # :151
    n = (1 - z) * h + z * ht                                                         # :153
    n = F.layer_norm(n, (n.shape[-1],), p["gamma"], p["beta"], eps)                  # :154
    return n + h                                                                     # :155


def global_sum_pool(h, atom_ids):
    return (h * (atom_ids > 0).to(h.dtype)[..., None]).sum(dim=1)                    # :163-164


def encode(w, prefix, atom_ids, bond_ids, conn, dtype=torch.float32, pooled_only=False):
    atom_ids, bond_ids, conn = torch.as_tensor(atom_ids), torch.as_tensor(bond_ids), torch.as_tensor(conn)
    h = F.embedding(atom_ids.long(), _t(w["atom_embedding"], dtype))                 # train_viscosity.py:171
    be = F.embedding(bond_ids.long(), _t(w["bond_embedding"], dtype))                # :172
    i = 0
    while f"{prefix}_bmm_{i}/bond_transform" in w:                                   # :176
        g = f"{prefix}_gu_{i}"
        p = {"Wz": _t(w[f"{g}/dense_z/kernel"], dtype), "bz": _t(w[f"{g}/dense_z/bias"], dtype),
             "Wr": _t(w[f"{g}/dense_r/kernel"], dtype), "br": _t(w[f"{g}/dense_r/bias"], dtype),
             "Wh": _t(w[f"{g}/dense_h/kernel"], dtype), "bh": _t(w[f"{g}/dense_h/bias"], dtype),
             "gamma": _t(w[f"{g}/layernorm/gamma"], dtype), "beta": _t(w[f"{g}/layernorm/beta"], dtype)}
        m = bond_matrix_message(h, be, conn, _t(w[f"{prefix}_bmm_{i}/bond_transform"], dtype))  # :178
        agg = reduce_messages(m, conn[:, :, 1], h.shape[1])                          # :182
        h = gated_update(h, agg, p)                                                  # :184
        i += 1
    pooled = global_sum_pool(h, atom_ids)                                            # :187
    if pooled_only:
        return pooled
    return torch.relu(pooled @ _t(w[f"{prefix}_fp/kernel"], dtype) + _t(w[f"{prefix}_fp/bias"], dtype))  # :189


def viscosity_forward(w, inputs, dtype=torch.float32):
    fc = encode(w, "cat", inputs["cat_atom"], inputs["cat_bond"], inputs["cat_connectivity"], dtype)
    fa = encode(w, "an", inputs["an_atom"], inputs["an_bond"], inputs["an_connectivity"], dtype)
    cp = torch.relu(fc @ _t(w["cat_proj/kernel"], dtype) + _t(w["cat_proj/bias"], dtype))     # :197
    ap = torch.relu(fa @ _t(w["an_proj/kernel"], dtype) + _t(w["an_proj/bias"], dtype))       # :198
    vp = (cp + ap) @ _t(w["visc_params/kernel"], dtype) + _t(w["visc_params/bias"], dtype)    # :201-204
    A = vp[:, 0:1]
    Bp = torch.clamp(F.softplus(vp[:, 1:2], threshold=1e9), 0.0, 20.0)
    Cp = torch.clamp(F.softplus(vp[:, 2:3], threshold=1e9), 0.1, 50.0)
    T = _t(inputs["temperature"], dtype) / 100.0
    return A + Bp / (T + Cp + 1e-6)


def melting_point_forward(w, inputs, dtype=torch.float32):
    """train_melting_point.py:146-208: K = D*D bond states, Add() mixing, Dense(relu) -> Dense(1)."""
    fc = encode(w, "cat", inputs["cat_atom"], inputs["cat_bond"], inputs["cat_connectivity"], dtype)
    fa = encode(w, "an", inputs["an_atom"], inputs["an_bond"], inputs["an_connectivity"], dtype)
    cp = torch.relu(fc @ _t(w["cat_proj/kernel"], dtype) + _t(w["cat_proj/bias"], dtype))     # :192
    ap = torch.relu(fa @ _t(w["an_proj/kernel"], dtype) + _t(w["an_proj/bias"], dtype))       # :193
    x = torch.relu((cp + ap) @ _t(w["mp_hidden/kernel"], dtype) + _t(w["mp_hidden/bias"], dtype))  # :197
    return x @ _t(w["mp_out/kernel"], dtype) + _t(w["mp_out/bias"], dtype)                    # :198


def pooled_pair(w, inputs, dtype=torch.float32):
    """The hot path only: both ions' GlobalSumPool outputs."""
    return (encode(w, "cat", inputs["cat_atom"], inputs["cat_bond"], inputs["cat_connectivity"], dtype, True),
            encode(w, "an", inputs["an_atom"], inputs["an_bond"], inputs["an_connectivity"], dtype, True))
