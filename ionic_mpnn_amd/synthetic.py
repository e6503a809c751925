"""Synthetic padded cation/anion graph batches.

The reference's datasets (viscosity_id_data.pkl, mp_id_data.pkl, vocab.pkl) are git-ignored
and absent, so every measured configuration runs on graphs drawn here:

  per ion: n_atoms ~ U{min_atoms..max_atoms}; a random spanning tree (n-1 bonds) plus
  U{0..1} ring closures, capped so that 2*n_bonds <= E; both directions of a bond are
  emitted adjacently as (u,v),(v,u) with **0-based** atom indices, exactly like
  src/featurize.py:54-63 - so the reference's "index 0 is padding" quirk
  (models/layers.py:74,114) is exercised; atom_ids ~ U{1..Va-1}, bond_ids ~ U{1..Vb-1}
  (the +1 shift of train_viscosity.py:255-262 already applied); zero padding to N, E.

Va=124 / Vb=72 are stand-ins (README.md:175-176 plus the padding id).
"""
from __future__ import annotations

import numpy as np

DEFAULT_VA = 124
DEFAULT_VB = 72


def _one_ion(rng, B, N, E, Va, Vb, min_atoms, max_atoms):
    n = rng.integers(min_atoms, max_atoms + 1, size=B)
    col = np.arange(N)
    # spanning tree: parent[v] uniform in [0, v)
    par = np.floor(rng.random((B, N)) * col[None, :]).astype(np.int64)
    atom_ids = rng.integers(1, Va, size=(B, N))
    atom_ids = np.where(col[None, :] < n[:, None], atom_ids, 0).astype(np.int32)

    nb_slots = E // 2
    bu = np.zeros((B, nb_slots), dtype=np.int64)
    bv = np.zeros((B, nb_slots), dtype=np.int64)
    bvalid = np.zeros((B, nb_slots), dtype=bool)
    jt = min(N - 1, nb_slots)
    v = np.arange(1, jt + 1)
    bu[:, :jt] = par[:, 1:jt + 1]
    bv[:, :jt] = v[None, :]
    bvalid[:, :jt] = v[None, :] < n[:, None]

    # 0/1 ring closure in the slot right after the tree bonds
    flag = rng.integers(0, 2, size=B).astype(bool)
    cu = np.floor(rng.random(B) * n).astype(np.int64)
    cv = np.floor(rng.random(B) * n).astype(np.int64)
    rows = np.arange(B)
    ok = flag & (cu != cv) & (par[rows, cv] != cu) & (par[rows, cu] != cv) & (n - 1 < nb_slots)
    slot = np.minimum(n - 1, nb_slots - 1)
    bu[rows[ok], slot[ok]] = cu[ok]
    bv[rows[ok], slot[ok]] = cv[ok]
    bvalid[rows[ok], slot[ok]] = True

    bid = rng.integers(1, Vb, size=(B, nb_slots))
    bid = np.where(bvalid, bid, 0)
    bu = np.where(bvalid, bu, 0)
    bv = np.where(bvalid, bv, 0)

    conn = np.zeros((B, E, 2), dtype=np.int32)
    bond = np.zeros((B, E), dtype=np.int32)
    conn[:, 0:2 * nb_slots:2, 0] = bu
    conn[:, 0:2 * nb_slots:2, 1] = bv
    conn[:, 1:2 * nb_slots:2, 0] = bv
    conn[:, 1:2 * nb_slots:2, 1] = bu
    bond[:, 0:2 * nb_slots:2] = bid
    bond[:, 1:2 * nb_slots:2] = bid
    return atom_ids, bond, conn


def make_batch(batch, max_atoms=40, max_edges=80, atom_vocab_size=DEFAULT_VA,
               bond_vocab_size=DEFAULT_VB, min_atoms=8, seed=0, with_temperature=True):
    """Returns the reference's 7 named model inputs (train_viscosity.py:150-160,306-314) as
    numpy arrays: int32 ids / connectivity, float32 temperature (B,1)."""
    rng = np.random.default_rng(seed)
    min_atoms = min(min_atoms, max_atoms)
    out = {}
    for p in ("cat", "an"):
        a, b, c = _one_ion(rng, batch, max_atoms, max_edges, atom_vocab_size, bond_vocab_size,
                           min_atoms, max_atoms)
        out[f"{p}_atom"], out[f"{p}_bond"], out[f"{p}_connectivity"] = a, b, c
    if with_temperature:
        out["temperature"] = rng.uniform(253.0, 393.0, size=(batch, 1)).astype(np.float32)
    return out


def _one_ion_explicit_h(rng, B, N, E, Va, Vb, min_atoms, max_atoms, max_degree=4, bond_type_probs=None):
    """Explicit-hydrogen-like molecules as the reference's trainers hand them to the model: featurize.py:45,54-63
    adds hydrogens and lists every bond in both directions; preprocess_edges_and_bonds (utils/mp_utils.py:18-45,
    train_viscosity.py:76-110) then follows every listed (src, tgt) with its reverse and repeats the bond id - FOUR
    edge slots per bond, (u,v),(v,u),(v,u),(u,v) - and pads to E = 2 * max(len(edge_indices)) = 4 * max_bonds slots
    (train_viscosity.py:288-289).  Skeleton: a random tree with atom degree <= max_degree plus up to two ring closures,
    so in-degrees stay <= 2 * max_degree."""
    max_atoms = min(max_atoms, N, E // 4)  # n - 1 tree bonds + rings must fit the slots
    n = rng.integers(min(min_atoms, max_atoms), max_atoms + 1, size=B)
    atom_ids = np.where(np.arange(N)[None, :] < n[:, None], rng.integers(1, Va, size=(B, N)), 0).astype(np.int32)
    deg = np.zeros((B, N), dtype=np.int64)
    par = np.zeros((B, N), dtype=np.int64)
    rows = np.arange(B)
    for v in range(1, N):
        u = np.floor(rng.random(B) * v).astype(np.int64)
        for _ in range(8):  # re-draw parents that are saturated
            full = deg[rows, u] >= max_degree
            if not full.any():
                break
            u = np.where(full, np.floor(rng.random(B) * v).astype(np.int64), u)
        full = deg[rows, u] >= max_degree
        if full.any():  # first unsaturated earlier atom (a tree of v atoms always has one)
            first = np.argmax(deg[:, :v] < max_degree, axis=1)
            u = np.where(full, first, u)
        live = v < n
        par[:, v] = u
        deg[rows[live], u[live]] += 1
        deg[rows[live], v] += 1
    nb_slots = E // 4
    bu = np.zeros((B, nb_slots), dtype=np.int64)
    bv = np.zeros((B, nb_slots), dtype=np.int64)
    bvalid = np.zeros((B, nb_slots), dtype=bool)
    jt = min(N - 1, nb_slots)
    vv = np.arange(1, jt + 1)
    bu[:, :jt] = par[:, 1:jt + 1]
    bv[:, :jt] = vv[None, :]
    bvalid[:, :jt] = vv[None, :] < n[:, None]
    for ring in range(2):  # ring closures between unsaturated atoms, in the slots after the tree bonds
        cu = np.floor(rng.random(B) * n).astype(np.int64)
        cv = np.floor(rng.random(B) * n).astype(np.int64)
        slot = n - 1 + ring
        ok = (cu != cv) & (par[rows, cv] != cu) & (par[rows, cu] != cv) & (slot < nb_slots) \
            & (deg[rows, cu] < max_degree) & (deg[rows, cv] < max_degree)
        slot = np.minimum(slot, nb_slots - 1)
        bu[rows[ok], slot[ok]] = cu[ok]
        bv[rows[ok], slot[ok]] = cv[ok]
        bvalid[rows[ok], slot[ok]] = True
        deg[rows[ok], cu[ok]] += 1
        deg[rows[ok], cv[ok]] += 1
    if bond_type_probs is None:
        draw = rng.integers(1, Vb, size=(B, nb_slots))
    else:  # a few bond types with skewed frequencies, as real molecules have (ids 1 .. len(probs))
        pr = np.asarray(bond_type_probs, dtype=np.float64)
        draw = 1 + rng.choice(len(pr), size=(B, nb_slots), p=pr / pr.sum())
    bid = np.where(bvalid, draw, 0)
    bu = np.where(bvalid, bu, 0)
    bv = np.where(bvalid, bv, 0)
    # the ring slot may sit beyond a gap-free prefix only when a ring was refused: compact nothing, the reference's
    # lists are gap-free but padding slots in the middle are legal inputs too ([0,0] / bond 0 = masked, layers.py:114)
    conn = np.zeros((B, E, 2), dtype=np.int32)
    bond = np.zeros((B, E), dtype=np.int32)
    for k, (s_, t_) in enumerate(((bu, bv), (bv, bu), (bv, bu), (bu, bv))):
        conn[:, k:4 * nb_slots:4, 0] = s_
        conn[:, k:4 * nb_slots:4, 1] = t_
        bond[:, k:4 * nb_slots:4] = bid
    return atom_ids, bond, conn


def make_explicit_h_batch(batch, max_atoms=160, max_edges=640, atom_vocab_size=DEFAULT_VA, bond_vocab_size=DEFAULT_VB,
                          min_atoms=20, seed=0, with_temperature=True, bond_type_probs=None):
    """As make_batch, for the padded shapes of the reference's real (explicit-hydrogen) data sets: up to `max_atoms`
    atoms of degree <= 4, every bond in four edge slots, E = 4 * max_bonds (see _one_ion_explicit_h).  Bond ids are
    uniform over the vocabulary unless `bond_type_probs` gives the frequencies of a few types (ids 1, 2, ...)."""
    rng = np.random.default_rng(seed)
    out = {}
    for p in ("cat", "an"):
        a, b, c = _one_ion_explicit_h(rng, batch, max_atoms, max_edges, atom_vocab_size, bond_vocab_size, min_atoms,
                                      max_atoms, bond_type_probs=bond_type_probs)
        out[f"{p}_atom"], out[f"{p}_bond"], out[f"{p}_connectivity"] = a, b, c
    if with_temperature:
        out["temperature"] = rng.uniform(253.0, 393.0, size=(batch, 1)).astype(np.float32)
    return out


def make_id_records(num, seed=0, min_atoms=3, max_atoms=12, atom_vocab=20, bond_vocab=6, kind="viscosity"):
    """Synthetic ``*_id_data.pkl`` records in the schema of src/dataset.py:15-20,51-62
    (0-based ids, featurize-style bidirectional edge list, python lists)."""
    rng = np.random.default_rng(seed)
    recs = []
    for r in range(num):
        rec = {"pair_id": f"pair_{r}"}
        for ion in ("cation", "anion"):
            n = int(rng.integers(min_atoms, max_atoms + 1))
            edges, bids = [], []
            for v in range(1, n):
                u = int(rng.integers(0, v))
                t = int(rng.integers(0, bond_vocab))
                edges += [(u, v), (v, u)]
                bids += [t, t]
            rec[ion] = {
                "atom_ids": [int(x) for x in rng.integers(0, atom_vocab, size=n)],
                "bond_ids": bids,
                "edge_indices": edges,
                "num_atoms": n,
            }
        if kind == "viscosity":
            rec["T"] = float(rng.uniform(253.0, 393.0))
            rec["log_eta"] = float(rng.normal(4.0, 1.0))
        else:
            rec["mp"] = float(rng.uniform(200.0, 500.0))
        recs.append(rec)
    vocab = {"atom_vocab": {}, "bond_vocab": {}, "atom_vocab_size": atom_vocab, "bond_vocab_size": bond_vocab}
    return recs, vocab
