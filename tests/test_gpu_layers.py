"""GPU parity: each drop-in layer (one libimpnn launch each) against the numpy oracle, on the
golden fixtures and on seeded random / edge-case inputs.  Tolerance: 1e-5 relative to the tensor's
scale (BASELINE.json north_star); integer-indexed pure copies/sums are checked bit-exact."""
import numpy as np
import pytest
import torch

from conftest import assert_close, load_case
from ionic_mpnn_amd import layers as L
from ionic_mpnn_amd import ops
from oracle import mpnn_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def f32(a):
    return np.asarray(a, dtype=np.float32)


CASES = ["tiny_viscosity", "config2_b8", "config2_perturbed_b6", "tiny_melting_point", "wide_d64_b5"]


@pytest.mark.parametrize("name", CASES)
def test_embedding_bit_exact(name):
    _, inp, w, outs = load_case(name)
    out = ops.embed_gather(dev(inp["cat_atom"]), dev(w["atom_embedding"])).cpu().numpy()
    np.testing.assert_array_equal(out, w["atom_embedding"][inp["cat_atom"]])      # integer indexing: bit-exact
    out = ops.embed_gather(dev(inp["an_bond"]), dev(w["bond_embedding"])).cpu().numpy()
    np.testing.assert_array_equal(out, w["bond_embedding"][inp["an_bond"]])


@pytest.mark.parametrize("name", CASES)
def test_bond_matrix_message_vs_golden(name):
    _, inp, w, outs = load_case(name)
    for p in ("cat", "an"):
        for i in range(O.num_steps_of(w)):
            h = f32(outs[f"{p}/h{i}"])
            bs = f32(outs[f"{p}/bond_emb"])
            W = w[f"{p}_bmm_{i}/bond_transform"]
            conn = inp[f"{p}_connectivity"]
            ref = O.bond_matrix_message(h.astype(np.float64), bs.astype(np.float64), conn, W.astype(np.float64))
            m = ops.bmm_message(dev(h), dev(bs), dev(conn), dev(W)).cpu().numpy()
            assert_close(m, ref, what=f"{name} {p} m{i}")
            # masked rows are exactly zero (models/layers.py:114-115)
            invalid = ~((conn[..., 0] > 0) & (conn[..., 1] > 0))
            assert (m[invalid] == 0).all()
            # per-bond-type schedule gives the same messages
            mats = ops.bond_type_matrices(dev(w["bond_embedding"]), dev(W))
            assert_close(mats.cpu().numpy(), np.tensordot(w["bond_embedding"].astype(np.float64), W.astype(np.float64), 1),
                         what="type matrices")
            mt = ops.bmm_message_typed(dev(h), dev(inp[f"{p}_bond"]), dev(conn), mats).cpu().numpy()
            assert_close(mt, ref, what=f"{name} {p} typed m{i}")


@pytest.mark.parametrize("name", CASES)
def test_reduce_bitwise_vs_sequential_scatter(name):
    _, inp, w, outs = load_case(name)
    for p in ("cat", "an"):
        m = f32(outs[f"{p}/m0"])
        conn = inp[f"{p}_connectivity"]
        ref32 = O.reduce_messages(m, conn[:, :, 1], np.zeros((m.shape[0], inp[f"{p}_atom"].shape[1], 1)))
        cd = dev(conn)
        agg_view = ops.reduce_scatter_add(dev(m), cd[:, :, 1], inp[f"{p}_atom"].shape[1]).cpu().numpy()   # strided view
        agg_cont = ops.reduce_scatter_add(dev(m), cd[:, :, 1].contiguous(), inp[f"{p}_atom"].shape[1]).cpu().numpy()
        # edge-slot-order accumulation == np.add.at in fp32, bit for bit
        np.testing.assert_array_equal(agg_view, ref32)
        np.testing.assert_array_equal(agg_cont, ref32)


@pytest.mark.parametrize("name", CASES)
def test_gated_update_vs_golden(name):
    _, inp, w, outs = load_case(name)
    for p in ("cat", "an"):
        for i in range(O.num_steps_of(w)):
            sp = O.step_params(w, p, i, np.float64)
            h, agg = f32(outs[f"{p}/h{i}"]), f32(outs[f"{p}/agg{i}"])
            ref = O.gated_update(h.astype(np.float64), agg.astype(np.float64), sp)
            g = {k: dev(f32(v)) for k, v in sp.items()}
            out = ops.gated_update(dev(h), dev(agg), g["Wz"], g["bz"], g["Wr"], g["br"], g["Wh"], g["bh"], g["gamma"],
                                   g["beta"]).cpu().numpy()
            assert_close(out, ref, what=f"{name} {p} h{i + 1}")


@pytest.mark.parametrize("name", CASES)
def test_global_sum_pool_vs_golden(name):
    _, inp, w, outs = load_case(name)
    S = O.num_steps_of(w)
    for p in ("cat", "an"):
        h = f32(outs[f"{p}/h{S}"])
        out = ops.global_sum_pool(dev(h), dev(inp[f"{p}_atom"])).cpu().numpy()
        assert_close(out, O.global_sum_pool(h.astype(np.float64), inp[f"{p}_atom"]), what=f"{name} {p} pooled")


def test_fused_message_reduce_orphan_signature():
    _, inp, w, outs = load_case("config2_b8")
    h, bs, conn = f32(outs["cat/h1"]), f32(outs["cat/bond_emb"]), inp["cat_connectivity"]
    W = w["cat_bmm_1/bond_transform"]
    ref = O.bond_matrix_message_fused(h.astype(np.float64), bs.astype(np.float64), conn,
                                      W.reshape(8, -1).astype(np.float64))
    agg = ops.bmm_fused(dev(h), dev(bs), dev(conn), dev(W.reshape(8, -1))).cpu().numpy()
    assert agg.shape == h.shape
    assert_close(agg, ref, what="bmm_fused")
    lyr = L.BondMatrixMessage(32, 8, fused=True, device=torch.device(DEV))
    lyr.build(None); lyr.built = True
    lyr.set_weights([W])
    assert_close(lyr([dev(h), dev(bs), dev(conn)]).cpu().numpy(), ref, what="BondMatrixMessage(fused=True)")


def test_layer_objects_chain_like_encode():
    """The reference's encode() loop body (train_viscosity.py:176-184) through the layer classes."""
    _, inp, w, outs = load_case("tiny_viscosity")
    d = torch.device(DEV)
    h = dev(f32(outs["cat/h0"]))
    bond = dev(f32(outs["cat/bond_emb"]))
    conn = dev(inp["cat_connectivity"])
    for i in range(2):
        bmm = L.BondMatrixMessage(8, 4, name=f"cat_bmm_{i}", device=d)
        red = L.Reduce(name=f"cat_reduce_{i}", device=d)
        gu = L.GatedUpdate(8, device=d)
        m = bmm([h, bond, conn])                        # builds on first call
        bmm.set_weights([w[f"cat_bmm_{i}/bond_transform"]])
        m = bmm([h, bond, conn])
        agg = red([m, conn[:, :, 1], h])
        gu([h, agg])
        gu.set_weights([w[f"cat_gu_{i}/{n}"] for n in gu.weight_names()])
        h = gu([h, agg], training=False)
        assert_close(m.cpu().numpy(), outs[f"cat/m{i}"], what=f"m{i}")
        assert_close(agg.cpu().numpy(), outs[f"cat/agg{i}"], what=f"agg{i}")
        assert_close(h.cpu().numpy(), outs[f"cat/h{i + 1}"], what=f"h{i + 1}")
    pooled = L.GlobalSumPool(device=d)([h, dev(inp["cat_atom"])])
    assert_close(pooled.cpu().numpy(), outs["cat/pooled"], what="pooled")


def test_type_matrix_cache_follows_weight_changes():
    """Embedding(lazy=True) -> BondMatrixMessage keeps A[v] = sum_k Tb[v,k] W[k] between calls; any in-place change of
    the bond table or of bond_transform must be seen by the next call."""
    _, inp, w, outs = load_case("tiny_viscosity")
    d = torch.device(DEV)
    h, conn, bond_ids = dev(f32(outs["cat/h0"])), dev(inp["cat_connectivity"]), dev(inp["cat_bond"])
    emb = L.Embedding(w["bond_embedding"].shape[0], 4, lazy=True, device=d)
    emb(bond_ids)
    bmm = L.BondMatrixMessage(8, 4, name="cat_bmm_0", device=d)
    first = bmm([h, emb(bond_ids), conn]).clone()          # random initial weights
    emb.set_weights([w["bond_embedding"]])
    bmm.set_weights([w["cat_bmm_0/bond_transform"]])
    m = bmm([h, emb(bond_ids), conn])
    assert not torch.equal(m, first)
    assert_close(m.cpu().numpy(), outs["cat/m0"], what="m0 after set_weights")
    assert torch.equal(bmm([h, emb(bond_ids), conn]), m)   # served from the cache
    with torch.no_grad():
        bmm.bond_transform.mul_(2.0)
    assert_close(bmm([h, emb(bond_ids), conn]).cpu().numpy(), 2.0 * outs["cat/m0"], what="m0 after in-place scale")


# ------------------------------------------------------------------ edge cases
def test_empty_and_degenerate_shapes():
    assert ops.embed_gather(torch.zeros(0, 5, dtype=torch.int32, device=DEV), dev(np.ones((3, 4), np.float32))).shape == (0, 5, 4)
    h = dev(np.ones((2, 3, 4), np.float32))
    e0 = ops.bmm_message(h, torch.zeros(2, 0, 2, device=DEV), torch.zeros(2, 0, 2, dtype=torch.int32, device=DEV),
                         dev(np.ones((2, 4, 4), np.float32)))
    assert e0.shape == (2, 0, 4)
    agg = ops.reduce_scatter_add(e0, torch.zeros(2, 0, dtype=torch.int32, device=DEV), 3)
    assert agg.shape == (2, 3, 4) and float(agg.abs().max()) == 0.0
    # all-padding molecule: everything masked
    ids = torch.zeros(1, 3, dtype=torch.int32, device=DEV)
    assert float(ops.global_sum_pool(h[:1], ids).abs().max()) == 0.0


def test_out_of_range_indices_never_fault_and_debug_mode_raises():
    rng = np.random.default_rng(0)
    h = rng.normal(size=(2, 5, 8)).astype(np.float32)
    bs = rng.normal(size=(2, 6, 3)).astype(np.float32)
    W = rng.normal(size=(3, 8, 8)).astype(np.float32)
    conn = np.array([[[1, 2], [7, 1], [2, -3], [4, 9], [1, 1], [0, 0]]] * 2, dtype=np.int32)
    m = ops.bmm_message(dev(h), dev(bs), dev(conn), dev(W)).cpu().numpy()
    clean = conn.copy()
    bad = (conn < 0).any(-1) | (conn >= 5).any(-1)
    clean[bad] = 0
    assert_close(m, O.bond_matrix_message(h.astype(np.float64), bs.astype(np.float64), clean, W.astype(np.float64)))
    agg = ops.reduce_scatter_add(dev(m), dev(conn[:, :, 1].copy()), 5).cpu().numpy()
    assert np.isfinite(agg).all()
    e = ops.embed_gather(dev(np.array([[0, 4, 5, -1]], np.int32)), dev(np.ones((5, 2), np.float32))).cpu().numpy()
    np.testing.assert_array_equal(e[0], [[1, 1], [1, 1], [0, 0], [0, 0]])
    ops.DEBUG_VALIDATE = True
    try:
        with pytest.raises(ValueError):                      # tf-CPU raises here (models/layers.py:106)
            ops.bmm_message(dev(h), dev(bs), dev(conn), dev(W))
    finally:
        ops.DEBUG_VALIDATE = False


@pytest.mark.parametrize("D,K,N,E,B", [(32, 8, 40, 80, 33), (16, 3, 7, 10, 5), (48, 2, 9, 30, 3), (128, 8, 20, 40, 4),
                                       (8, 64, 10, 20, 6)])
def test_random_shapes_all_layers(D, K, N, E, B):
    rng = np.random.default_rng(D * 1000 + K)
    h = rng.normal(size=(B, N, D)).astype(np.float32)
    bs = rng.normal(size=(B, E, K)).astype(np.float32)
    W = (rng.normal(size=(K, D, D)) / np.sqrt(D * K)).astype(np.float32)
    conn = rng.integers(0, N, size=(B, E, 2)).astype(np.int32)
    conn[:, E // 2:] = 0
    ids = rng.integers(0, 3, size=(B, N)).astype(np.int32)
    m_ref = O.bond_matrix_message(h.astype(np.float64), bs.astype(np.float64), conn, W.astype(np.float64))
    m = ops.bmm_message(dev(h), dev(bs), dev(conn), dev(W))
    assert_close(m.cpu().numpy(), m_ref, what="m")
    agg = ops.reduce_scatter_add(m, dev(conn)[:, :, 1], N)
    np.testing.assert_array_equal(agg.cpu().numpy(), O.reduce_messages(m.cpu().numpy(), conn[:, :, 1], h))
    assert_close(ops.bmm_fused(dev(h), dev(bs), dev(conn), dev(W)).cpu().numpy(), O.reduce_messages(m_ref, conn[:, :, 1], h),
                 what="fused agg")
    p = {k: (rng.normal(size=(2 * D, D)) / np.sqrt(2 * D)).astype(np.float32) for k in ("Wz", "Wr", "Wh")}
    p.update({k: rng.normal(size=D).astype(np.float32) * 0.1 for k in ("bz", "br", "bh", "beta")})
    p["gamma"] = rng.uniform(0.5, 1.5, size=D).astype(np.float32)
    a = agg.cpu().numpy()
    ref = O.gated_update(h.astype(np.float64), a.astype(np.float64), {k: v.astype(np.float64) for k, v in p.items()})
    g = {k: dev(v) for k, v in p.items()}
    out = ops.gated_update(dev(h), agg, g["Wz"], g["bz"], g["Wr"], g["br"], g["Wh"], g["bh"], g["gamma"], g["beta"])
    assert_close(out.cpu().numpy(), ref, what="gated_update")
    assert_close(ops.global_sum_pool(out, dev(ids)).cpu().numpy(), O.global_sum_pool(ref, ids), what="pool")


@pytest.mark.parametrize("D,N,E", [(32, 64, 256), (64, 20, 300), (8, 33, 77), (128, 9, 513)])
def test_reduce_small_batch_kernel_is_bitwise_the_large_batch_kernel(D, N, E):
    """Batches under 2048 molecules take reduce_scatter_small_kernel (target-partitioned thread groups, messages in
    flight 16 at a time); both kernels form every (target, column) sum in edge-slot order, so the first molecules of
    a 2100-molecule call equal a small call on the same molecules bit for bit - and the sequential numpy scatter."""
    rng = np.random.default_rng(D + E)
    B, Bs = 2100, 37
    m = torch.from_numpy(rng.normal(size=(B, E, D)).astype(np.float32)).to(DEV)
    tgt = rng.integers(-1, N + 2, size=(B, E)).astype(np.int32)   # includes 0 (padding), negative and >= N (skipped)
    tgt[:, rng.integers(0, E, size=E // 3)] = 3                    # a crowded target row
    big = ops.reduce_scatter_add(m, dev(tgt), N)
    small = ops.reduce_scatter_add(m[:Bs].contiguous(), dev(tgt[:Bs]), N)
    assert torch.equal(big[:Bs], small)
    ref = np.zeros((Bs, N, D), np.float32)
    mh = m[:Bs].cpu().numpy()
    for b in range(Bs):
        for e in range(E):
            t = tgt[b, e]
            if 0 < t < N:
                ref[b, t] += mh[b, e]
    np.testing.assert_array_equal(small.cpu().numpy(), ref)


@pytest.mark.parametrize("D", [32, 64, 128])
def test_gated_update_on_kept_rows_only(D):
    """impnn_kept_rows / impnn_row_index_fill / impnn_gated_update_rows (the model's layered path at wide states):
    the row list is exactly the encoder's kept rows, the listed rows equal the full GatedUpdate bit for bit, every
    other row of the output is left alone."""
    import numpy as np
    from ionic_mpnn_amd import synthetic
    B, N, E, Vb = 37, 24, 40, 9
    inp = synthetic.make_batch(B, max_atoms=N, max_edges=E, atom_vocab_size=12, bond_vocab_size=Vb, min_atoms=3, seed=D)
    ids, bond, conn = (torch.from_numpy(inp[k]).to(DEV) for k in ("cat_atom", "cat_bond", "cat_connectivity"))
    idx, cnt = ops.kept_row_index(ids, bond, conn, Vb)
    # host model of the rule: r_b = 1 + max(last n with id > 0, largest index on a valid edge)
    want = []
    for b in range(B):
        r = int(np.max(np.nonzero(inp["cat_atom"][b])[0], initial=-1)) + 1
        c = inp["cat_connectivity"][b]
        ok = (c[:, 0] > 0) & (c[:, 1] > 0)
        if ok.any():
            r = max(r, int(c[ok].max()) + 1)
        want += [b * N + n for n in range(r)]
    n = int(cnt.item())
    assert n == len(want) and idx[:n].cpu().tolist() == want
    g = torch.Generator().manual_seed(D)
    h, agg = torch.randn(B, N, D, generator=g).to(DEV), torch.randn(B, N, D, generator=g).to(DEV)
    W = [(torch.randn(2 * D, D, generator=g) / (2 * D) ** 0.5).to(DEV), torch.randn(D, generator=g).to(DEV) * 0.1,
         (torch.randn(2 * D, D, generator=g) / (2 * D) ** 0.5).to(DEV), torch.randn(D, generator=g).to(DEV) * 0.1,
         (torch.randn(2 * D, D, generator=g) / (2 * D) ** 0.5).to(DEV), torch.randn(D, generator=g).to(DEV) * 0.1,
         torch.rand(D, generator=g).to(DEV) + 0.5, torch.randn(D, generator=g).to(DEV) * 0.1]
    full = ops.gated_update(h, agg, *W)
    part = ops.gated_update(h, agg, *W, rows=(idx, cnt))
    keep = torch.zeros(B * N, dtype=torch.bool, device=DEV)
    keep[idx[:n].long()] = True
    assert torch.equal(part.reshape(B * N, D)[keep], full.reshape(B * N, D)[keep])
