"""CPU: the oracle against its committed golden vectors, hand-derived known answers, and an
independent second restatement (torch-CPU, reference op schedule)."""
import numpy as np
import pytest
import torch

from conftest import assert_close, load_case
from ionic_mpnn_amd import synthetic, weights
from oracle import mpnn_oracle as O
from oracle import torch_ref as TR

CASES = ["tiny_viscosity", "config2_b8", "config2_perturbed_b6", "tiny_melting_point", "wide_d64_b5"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden_fp64(name):
    kind, inputs, w, outs = load_case(name)
    trace = {}
    fwd = O.viscosity_forward if kind == "viscosity" else O.melting_point_forward
    y = fwd(w, inputs, np.float64, trace)
    np.testing.assert_allclose(y, outs["final"], rtol=1e-12, atol=1e-12)
    for k, v in trace.items():
        np.testing.assert_allclose(v, outs[k], rtol=1e-12, atol=1e-12, err_msg=k)


@pytest.mark.parametrize("name", CASES)
def test_oracle_fp32_within_tolerance_of_fp64(name):
    kind, inputs, w, outs = load_case(name)
    trace = {}
    fwd = O.viscosity_forward if kind == "viscosity" else O.melting_point_forward
    fwd(w, inputs, np.float32, trace)
    for k in ("cat/pooled", "an/pooled", "cat/fp", "mixed"):
        assert trace[k].dtype == np.float32
        assert_close(trace[k], outs[k], rel=1e-5, what=k)


@pytest.mark.parametrize("name", ["tiny_viscosity", "config2_b8", "config2_perturbed_b6", "wide_d64_b5"])
def test_torch_restatement_agrees_with_numpy_oracle(name):
    kind, inputs, w, outs = load_case(name)
    y = TR.viscosity_forward(w, inputs, torch.float64).numpy()
    assert_close(y, outs["final"], rel=1e-10, what="log_eta fp64")
    pc, pa = TR.pooled_pair(w, inputs, torch.float32)
    assert_close(pc.numpy(), outs["cat/pooled"], rel=1e-5, what="cat pooled fp32")
    assert_close(pa.numpy(), outs["an/pooled"], rel=1e-5, what="an pooled fp32")


# ---------------------------------------------------------------- hand-derived known answers
def _eye_transform(K, D):
    W = np.zeros((K, D, D))
    for k in range(K):
        W[k] = (k + 1) * np.eye(D)
    return W


def test_message_identity_transform_known_answer():
    # W[k] = (k+1) I  =>  m_e = (sum_k (k+1) bs[e,k]) * h[src_e]
    h = np.arange(1, 1 + 3 * 2, dtype=np.float64).reshape(1, 3, 2)          # atoms 0,1,2
    conn = np.array([[[1, 2], [2, 1], [0, 1], [1, 0], [0, 0]]], dtype=np.int32)
    bs = np.array([[[1.0, 0.0], [0.0, 1.0], [1.0, 1.0], [1.0, 1.0], [5.0, 5.0]]])
    m = O.bond_matrix_message(h, bs, conn, _eye_transform(2, 2))
    expect = np.zeros((1, 5, 2))
    expect[0, 0] = 1.0 * h[0, 1]          # src=1 -> tgt=2, coefficient 1
    expect[0, 1] = 2.0 * h[0, 2]          # src=2 -> tgt=1, coefficient 2
    # edges touching atom 0 and the [0,0] padding edge are zero: the index-0 quirk (layers.py:114-115)
    np.testing.assert_array_equal(m, expect)


def test_atom0_quirk_and_duplicate_accumulation():
    # chain 0-1-2 passed through the trainer's 4x expansion: each neighbour counts twice,
    # atom 0 neither sends nor receives
    conn, bond = O.preprocess_edges_and_bonds([[(0, 1), (1, 0), (1, 2), (2, 1)]], [[1, 1, 1, 1]], 4)
    h = np.array([[[10.0], [1.0], [100.0]]])
    bs = np.ones((1, 8, 1))
    m = O.bond_matrix_message(h, bs, conn, np.ones((1, 1, 1)))
    agg = O.reduce_messages(m, conn[:, :, 1], h)
    np.testing.assert_array_equal(agg[0, :, 0], [0.0, 2 * 100.0, 2 * 1.0])


def test_reduce_ignores_tgt0_and_accumulates_duplicates():
    m = np.arange(12, dtype=np.float64).reshape(1, 4, 3)
    tgt = np.array([[2, 0, 2, 1]], dtype=np.int32)
    agg = O.reduce_messages(m, tgt, np.zeros((1, 3, 3)))
    np.testing.assert_array_equal(agg[0], [[0, 0, 0], m[0, 3], m[0, 0] + m[0, 2]])


def test_out_of_range_indices_raise_like_tf_cpu():
    h = np.zeros((1, 3, 2))
    with pytest.raises(ValueError):
        O.bond_matrix_message(h, np.zeros((1, 1, 1)), np.array([[[3, 1]]], np.int32), np.zeros((1, 2, 2)))
    with pytest.raises(ValueError):
        O.reduce_messages(np.zeros((1, 1, 2)), np.array([[3]], np.int32), h)
    with pytest.raises(ValueError):
        O.embedding(np.array([5]), np.zeros((5, 2)))


def test_gated_update_known_answer_zero_weights():
    # zero kernels/biases: z = r = 0.5, h~ = 0, n = 0.5 h, out = LN(0.5 h) + h
    D = 4
    h = np.array([[[1.0, 2.0, 3.0, 6.0]]])
    p = {k: np.zeros((2 * D, D)) for k in ("Wz", "Wr", "Wh")}
    p.update({k: np.zeros(D) for k in ("bz", "br", "bh", "beta")})
    p["gamma"] = np.ones(D)
    out = O.gated_update(h, np.ones_like(h), p)
    n = 0.5 * h
    mu, var = n.mean(), n.var()
    np.testing.assert_allclose(out, (n - mu) / np.sqrt(var + 1e-3) + h, rtol=1e-14)


def test_global_sum_pool_masks_by_atom_id():
    h = np.arange(8, dtype=np.float64).reshape(1, 4, 2)
    ids = np.array([[3, 0, 7, 0]], dtype=np.int32)
    np.testing.assert_array_equal(O.global_sum_pool(h, ids), [[0 + 4, 1 + 5]])


def test_padded_atom_invariance():
    # growing N/E padding must not change the pooled fingerprint (padding atoms are masked at the
    # pool and no valid edge can name them)
    w = weights.init_weights("viscosity", 30, 9, atom_dim=8, bond_dim=4, num_steps=3, seed=3, perturb=True)
    a = synthetic.make_batch(5, max_atoms=9, max_edges=18, atom_vocab_size=30, bond_vocab_size=9, min_atoms=3, seed=4)
    fp_small = O.encode(w, "cat", a["cat_atom"], a["cat_bond"], a["cat_connectivity"], pooled_only=True)
    pad = lambda x, n, axis: np.concatenate(
        [x, np.zeros(x.shape[:axis] + (n,) + x.shape[axis + 1:], x.dtype)], axis=axis)
    fp_big = O.encode(w, "cat", pad(a["cat_atom"], 5, 1), pad(a["cat_bond"], 6, 1), pad(a["cat_connectivity"], 6, 1),
                      pooled_only=True)
    np.testing.assert_allclose(fp_big, fp_small, rtol=1e-13, atol=1e-13)


def test_batch_shard_concat_identical():
    kind, inputs, w, outs = load_case("config2_b8")
    lo = {k: v[:3] for k, v in inputs.items()}
    hi = {k: v[3:] for k, v in inputs.items()}
    y = np.concatenate([O.viscosity_forward(w, lo), O.viscosity_forward(w, hi)])
    # numpy's BLAS blocks differently per batch size: equal to rounding, not bitwise (the HIP path
    # is bitwise shard-invariant, tests/test_gpu_encoder.py)
    np.testing.assert_allclose(y, outs["final"], rtol=1e-13, atol=1e-13)


def test_edge_permutation_changes_only_rounding():
    kind, inputs, w, outs = load_case("tiny_viscosity")
    rng = np.random.default_rng(0)
    perm = rng.permutation(inputs["cat_bond"].shape[1])
    fp = O.encode(w, "cat", inputs["cat_atom"], inputs["cat_bond"][:, perm], inputs["cat_connectivity"][:, perm],
                  pooled_only=True)
    np.testing.assert_allclose(fp, outs["cat/pooled"], rtol=1e-12, atol=1e-12)


def test_synthetic_generator_contract():
    b = synthetic.make_batch(64, seed=0)
    for p in ("cat", "an"):
        ids, bond, conn = b[f"{p}_atom"], b[f"{p}_bond"], b[f"{p}_connectivity"]
        assert ids.dtype == bond.dtype == conn.dtype == np.int32
        n = (ids > 0).sum(1)
        assert n.min() >= 8 and n.max() <= 40
        assert ((ids > 0) == (np.arange(40)[None] < n[:, None])).all()      # ids are a prefix
        valid = bond > 0
        assert (conn[~valid] == 0).all()
        assert (conn[valid].max(axis=-1) < np.repeat(n, valid.sum(1))).all()  # edges name real atoms only
        # adjacent (u,v),(v,u) pairs with the same bond id
        assert (conn[:, 0::2, 0] == conn[:, 1::2, 1]).all() and (conn[:, 0::2, 1] == conn[:, 1::2, 0]).all()
        assert (bond[:, 0::2] == bond[:, 1::2]).all()
        nb = valid.sum(1) // 2
        assert ((nb == n - 1) | (nb == n)).all()                             # tree + 0/1 ring closure
        assert (conn[..., 0] == 0).any()                                      # atom 0 appears: quirk exercised
    again = synthetic.make_batch(64, seed=0)
    assert all((b[k] == again[k]).all() for k in b)
