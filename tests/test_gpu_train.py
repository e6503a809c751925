"""f4: backward kernels, optimizer step and fit().  Gradient oracle: torch-CPU fp64 autograd
through oracle/torch_ref.py (the reference's op schedule); tolerance 1e-4 of each tensor's scale (f32
kernels, sums of a few thousand terms).  Optimizer oracle: oracle/train_oracle.py."""
import numpy as np
import pytest
import torch

from ionic_mpnn_amd import model as MM, ops, synthetic, train, weights
from oracle import torch_ref as TR, train_oracle as TO

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(got, ref, tol=1e-4, what=""):
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    ref = ref.detach().cpu().double().numpy() if torch.is_tensor(ref) else np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = max(np.abs(ref).max(), 1e-12)
    err = np.abs(got - ref).max() / scale
    assert err <= tol, f"{what}: rel err {err:.3e} > {tol}"


def rand_graph(B, N, E, Vb, rng):
    conn = rng.integers(0, N, size=(B, E, 2)).astype(np.int32)
    conn[:, ::5, 0] = 0  # padding-like edges (masked: src == 0)
    bond = rng.integers(0, Vb, size=(B, E)).astype(np.int32)
    ids = rng.integers(0, 9, size=(B, N)).astype(np.int32)
    return conn, bond, ids


@pytest.mark.parametrize("D,K,B,E", [(8, 4, 5, 14), (32, 8, 5, 14), (64, 3, 5, 14), (128, 8, 40, 30), (64, 8, 90, 30)])
def test_message_reduce_backward(D, K, B, E):
    rng = np.random.default_rng(D)
    N, Vb = 9, 7
    conn, bond, _ = rand_graph(B, N, E, Vb, rng)
    h = rng.normal(size=(B, N, D)); tb = rng.normal(size=(Vb, K)); W = rng.normal(size=(K, D, D)) / np.sqrt(D)
    go = rng.normal(size=(B, N, D))
    # oracle (fp64)
    ho, tbo, Wo = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (h, tb, W))
    bs = torch.nn.functional.embedding(torch.tensor(bond).long(), tbo)
    m = TR.bond_matrix_message(ho, bs, torch.tensor(conn), Wo)
    agg = TR.reduce_messages(m, torch.tensor(conn)[:, :, 1], N)
    (agg * torch.tensor(go)).sum().backward()
    # HIP
    hg, tbg, Wg = (torch.tensor(a, dtype=torch.float32, device=DEV, requires_grad=True) for a in (h, tb, W))
    cg, bg = torch.tensor(conn, device=DEV), torch.tensor(bond, device=DEV)
    mats = ops.bond_type_matrices(tbg, Wg)
    mg = ops.bmm_message_typed(hg, bg, cg, mats)
    ag = ops.reduce_scatter_add(mg, cg[:, :, 1], N)
    close(ag, agg, 1e-5, "agg")
    (ag * torch.tensor(go, dtype=torch.float32, device=DEV)).sum().backward()
    close(hg.grad, ho.grad, what="dh")
    close(Wg.grad, Wo.grad, what="dW")
    close(tbg.grad, tbo.grad, what="dbond_table")
    # the two layers as ONE node (impnn_message_reduce_typed_bwd gathers dagg at the edge targets)
    from ionic_mpnn_amd import autograd
    hf, tbf, Wf = (torch.tensor(a, dtype=torch.float32, device=DEV, requires_grad=True) for a in (h, tb, W))
    af = autograd.MessageReduceTyped.apply(hf, bg, cg, ops.bond_type_matrices(tbf, Wf))
    assert torch.equal(af, ag)
    (af * torch.tensor(go, dtype=torch.float32, device=DEV)).sum().backward()
    close(hf.grad, ho.grad, what="dh (one node)")
    close(Wf.grad, Wo.grad, what="dW (one node)")
    close(tbf.grad, tbo.grad, what="dbond_table (one node)")


@pytest.mark.parametrize("D,Vb,from_agg,B", [(128, 12, True, 700), (64, 72, True, 700), (128, 3, False, 700), (128, 72, True, 32), (64, 9, True, 200)])
def test_message_backward_on_the_matrix_cores_equals_the_valu_kernel(D, Vb, from_agg, B):
    """Wide states take bmm_message_typed_bwd_mfma_kernel (csrc/train_kernels.hip): same gradients as the VALU kernel
    (IMPNN_MESSAGE_BWD=valu) on a batch whose type runs span several segments and several workgroup ranges - types
    change inside a workgroup's range, last segments of a type are partial.  (Both kernels add into dh / dA with
    float atomics: equal up to the order of f32 additions.)"""
    import os
    from ionic_mpnn_amd import _lib
    rng = np.random.default_rng(D + Vb)
    N, E = 40, 80
    conn, bond, _ = rand_graph(B, N, E, Vb, rng)
    h = torch.tensor(rng.normal(size=(B, N, D)), dtype=torch.float32, device=DEV)
    mats = torch.tensor(rng.normal(size=(Vb, D, D)) / np.sqrt(D), dtype=torch.float32, device=DEV)
    dm = torch.tensor(rng.normal(size=(B, N, D) if from_agg else (B, E, D)), dtype=torch.float32, device=DEV)
    cg, bg = torch.tensor(conn, device=DEV), torch.tensor(bond, device=DEV)
    lib = _lib.load()

    def run():
        dh = torch.zeros(B, N, D, device=DEV)
        dA = torch.zeros_like(mats)
        nb = int(lib.impnn_bmm_message_typed_bwd_workspace_bytes(B, E, Vb))
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        fn = lib.impnn_message_reduce_typed_bwd if from_agg else lib.impnn_bmm_message_typed_bwd
        _lib.check(fn(ops.ptr(h), ops.ptr(bg), ops.ptr(cg), ops.ptr(mats), ops.ptr(dm), ops.ptr(dh), ops.ptr(dA),
                      ops.ptr(ws), ws.numel(), B, N, E, D, Vb, 0, _lib.stream_ptr()))
        torch.cuda.synchronize()
        return dh, dA

    os.environ["IMPNN_MESSAGE_BWD"] = "mfma"
    try:
        dh1, dA1 = run()
        os.environ["IMPNN_MESSAGE_BWD"] = "valu"
        dh0, dA0 = run()
    finally:
        del os.environ["IMPNN_MESSAGE_BWD"]
    close(dh1, dh0.double().cpu(), 5e-6, "dh mfma vs valu")
    close(dA1, dA0.double().cpu(), 5e-6, "dA mfma vs valu")


@pytest.mark.parametrize("D,Vb,B,N,E", [(128, 12, 700, 40, 80), (64, 72, 700, 40, 80), (128, 72, 32, 40, 80), (128, 9, 40, 160, 640),
                                        (64, 5, 3000, 12, 30), (32, 72, 900, 40, 80), (16, 7, 50, 9, 14)])
def test_message_backward_through_the_edge_buffer(D, Vb, B, N, E):
    """impnn_message_reduce_typed_bwd_scratch: per-edge vectors into the (zero-rowed) message buffer, then slot-order sums
    at the source rows on top of what dh holds - the same dh as the atomics form up to f32 addition order, equal bits
    from run to run, masked edges' rows left zero."""
    from ionic_mpnn_amd import _lib
    rng = np.random.default_rng(D + Vb + B)
    conn, bond, _ = rand_graph(B, N, E, Vb, rng)
    h = torch.tensor(rng.normal(size=(B, N, D)), dtype=torch.float32, device=DEV)
    mats = torch.tensor(rng.normal(size=(Vb, D, D)) / np.sqrt(D), dtype=torch.float32, device=DEV)
    dagg = torch.tensor(rng.normal(size=(B, N, D)), dtype=torch.float32, device=DEV)
    dh_init = torch.tensor(rng.normal(size=(B, N, D)), dtype=torch.float32, device=DEV)
    cg, bg = torch.tensor(conn, device=DEV), torch.tensor(bond, device=DEV)
    lib = _lib.load()
    nb = int(lib.impnn_bmm_message_typed_bwd_workspace_bytes(B, E, Vb))
    # the buffer as a forward call leaves it: messages at valid edges, zero rows elsewhere
    m = ops.bmm_message_typed(h, bg, cg, mats)
    valid = (cg[:, :, 0] > 0) & (cg[:, :, 1] > 0) & (bg >= 0) & (bg < Vb)
    assert float(m[~valid].abs().max()) == 0.0 if (~valid).any() else True

    def run(scratch):
        dh, dA = dh_init.clone(), torch.zeros_like(mats)
        ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
        common = (ops.ptr(h), ops.ptr(bg), ops.ptr(cg), ops.ptr(mats), ops.ptr(dagg), ops.ptr(dh), ops.ptr(dA), ops.ptr(ws),
                  ws.numel())
        if scratch is None:
            _lib.check(lib.impnn_message_reduce_typed_bwd(*common, B, N, E, D, Vb, 0, _lib.stream_ptr()))
        else:
            _lib.check(lib.impnn_message_reduce_typed_bwd_scratch(*common, ops.ptr(scratch), B, N, E, D, Vb, 0,
                                                                  _lib.stream_ptr()))
        torch.cuda.synchronize()
        return dh, dA

    dh0, dA0 = run(None)
    buf = m.clone()
    dh1, dA1 = run(buf)
    dh2, _ = run(buf)          # the buffer is reused as it comes back (a training loop does)
    close(dh1, dh0.double().cpu(), 5e-6, "dh edge buffer vs atomics")
    close(dA1, dA0.double().cpu(), 5e-6, "dA edge buffer vs atomics")
    assert torch.equal(dh1, dh2)
    if (~valid).any():
        assert float(buf[~valid].abs().max()) == 0.0


@pytest.mark.parametrize("D,K,n,sinks", [(32, 8, 6, True), (8, 4, 3, False), (16, 5, 18, True), (64, 8, 3, True),
                                         (128, 8, 12, False), (32, 12, 4, True)])
def test_type_matrices_of_all_layers_in_one_node(D, K, n, sinks):
    """impnn_bond_type_matrices_multi[_bwd] against the per-layer entries: bitwise equal A_p, gradients equal to the
    per-layer sums (more than 16 layers take a second launch)."""
    from ionic_mpnn_amd import autograd
    rng = np.random.default_rng(D + n)
    Vb = 37
    tb0 = rng.normal(size=(Vb, K)).astype(np.float32)
    W0 = [(rng.normal(size=(K, D, D)) / np.sqrt(D)).astype(np.float32) for _ in range(n)]
    go = [torch.tensor(rng.normal(size=(Vb, D, D)).astype(np.float32), device=DEV) for _ in range(n)]

    def leaves():
        tb = torch.tensor(tb0, device=DEV, requires_grad=True)
        Ws = [torch.tensor(w, device=DEV, requires_grad=True) for w in W0]
        if sinks:
            for t in [tb] + Ws:
                t.grad = torch.full_like(t, 0.25)  # the kernels must ADD to what is there
        return tb, Ws

    tb1, W1 = leaves()
    ref = [ops.bond_type_matrices(tb1, w) for w in W1]
    sum((r * g).sum() for r, g in zip(ref, go)).backward()
    tb2, W2 = leaves()
    got = autograd.BondTypeMatricesAll.apply(tb2, *W2)
    for a, b in zip(got, ref):
        assert torch.equal(a, b)
    sum((r * g).sum() for r, g in zip(got, go)).backward()
    close(tb2.grad, tb1.grad, 1e-5, "dbond_table")
    for a, b in zip(W2, W1):
        close(a.grad, b.grad, 1e-6, "dW")


@pytest.mark.parametrize("D,rows", [(8, 37), (32, 301), (32, 70001), (64, 1500), (128, 19), (128, 1280), (128, 9000), (64, 8300)])
def test_gated_update_backward(D, rows):
    rng = np.random.default_rng(D + 1)
    names = ["Wz", "bz", "Wr", "br", "Wh", "bh", "gamma", "beta"]
    vals = {"Wz": rng.normal(size=(2 * D, D)) / np.sqrt(2 * D), "Wr": rng.normal(size=(2 * D, D)) / np.sqrt(2 * D),
            "Wh": rng.normal(size=(2 * D, D)) / np.sqrt(2 * D), "bz": rng.normal(size=D) * 0.1,
            "br": rng.normal(size=D) * 0.1, "bh": rng.normal(size=D) * 0.1, "gamma": 1 + 0.1 * rng.normal(size=D),
            "beta": 0.1 * rng.normal(size=D)}
    h, agg, go = rng.normal(size=(rows, D)), rng.normal(size=(rows, D)), rng.normal(size=(rows, D))
    po = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in vals.items()}
    ho, ao = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (h, agg))
    out = TR.gated_update(ho, ao, po)
    (out * torch.tensor(go)).sum().backward()
    pg = {k: torch.tensor(v, dtype=torch.float32, device=DEV, requires_grad=True) for k, v in vals.items()}
    hg, ag = (torch.tensor(a, dtype=torch.float32, device=DEV, requires_grad=True) for a in (h, agg))
    og = ops.gated_update(hg, ag, *[pg[k] for k in names])
    close(og, out, 1e-5, "out")
    (og * torch.tensor(go, dtype=torch.float32, device=DEV)).sum().backward()
    close(hg.grad, ho.grad, what="dh")
    close(ag.grad, ao.grad, what="dagg")
    for k in names:
        close(pg[k].grad, po[k].grad, what=f"d{k}")
    # parameter gradients go through fixed-order partial sums: bitwise reproducible
    g1 = pg["Wz"].grad.clone()
    for t in list(pg.values()) + [hg, ag]:
        t.grad = None
    og = ops.gated_update(hg, ag, *[pg[k] for k in names])
    (og * torch.tensor(go, dtype=torch.float32, device=DEV)).sum().backward()
    assert torch.equal(pg["Wz"].grad, g1)


@pytest.mark.parametrize("D,rows,keep", [(128, 1000, 0.6), (64, 5000, 0.3), (128, 70, 1.0), (64, 64, 0.0), (128, 12000, 0.7)])
def test_gated_update_backward_on_a_row_list(D, rows, keep):
    """impnn_gated_update_rows_bwd (the adjoint of GatedUpdate on the kept rows of an encode() loop) against the fp64
    oracle restricted to the listed rows: dh / dagg of the listed rows, parameter gradients = sums over them; rows
    outside the list are left untouched (dh) and the count lives on the device."""
    from ionic_mpnn_amd import autograd
    rng = np.random.default_rng(D + rows)
    names = ["Wz", "bz", "Wr", "br", "Wh", "bh", "gamma", "beta"]
    vals = {"Wz": rng.normal(size=(2 * D, D)) / np.sqrt(2 * D), "Wr": rng.normal(size=(2 * D, D)) / np.sqrt(2 * D),
            "Wh": rng.normal(size=(2 * D, D)) / np.sqrt(2 * D), "bz": rng.normal(size=D) * 0.1,
            "br": rng.normal(size=D) * 0.1, "bh": rng.normal(size=D) * 0.1, "gamma": 1 + 0.1 * rng.normal(size=D),
            "beta": 0.1 * rng.normal(size=D)}
    h, agg, go = rng.normal(size=(rows, D)), rng.normal(size=(rows, D)), rng.normal(size=(rows, D))
    sel = np.flatnonzero(rng.random(rows) < keep).astype(np.int32)
    n = len(sel)
    po = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in vals.items()}
    ho, ao = (torch.tensor(a[sel], dtype=torch.float64, requires_grad=True) for a in (h, agg))
    if n:
        (TR.gated_update(ho, ao, po) * torch.tensor(go[sel])).sum().backward()
    idx = torch.zeros(rows, dtype=torch.int32, device=DEV)
    idx[:n] = torch.tensor(sel, device=DEV)
    idx[n:] = -7                                  # entries past the count are never read
    cnt = torch.tensor([n], dtype=torch.int32, device=DEV)
    f32 = lambda a: torch.tensor(a, dtype=torch.float32, device=DEV)
    saved = (f32(h), f32(agg), *[f32(vals[k]) for k in names])
    dh, dagg, *dp = autograd._gated_update_backward(saved, 1e-3, f32(go), (idx, cnt))[:10]
    torch.cuda.synchronize()
    if n:
        close(dh[sel], ho.grad, what="dh of the listed rows")
        close(dagg[sel], ao.grad, what="dagg of the listed rows")
        for k, g in zip(names, dp):
            close(g, po[k].grad, what=f"d{k}")
    else:
        for g in dp:
            assert float(g.abs().max()) == 0.0
    rest = np.setdiff1d(np.arange(rows), sel)
    assert float(dh[rest].abs().max()) == 0.0 if len(rest) else True


@pytest.mark.parametrize("D,rows,keep", [(128, 1000, 0.6), (64, 5000, 0.3), (128, 70, 1.0), (64, 64, 0.0), (128, 12000, 0.7),
                                         (128, 1280, None), (64, 40, None), (32, 5000, None), (32, 7, None)])
def test_gated_update_backward_from_kept_activations(D, rows, keep):
    """impnn_gated_update_rows_train + impnn_gated_update_rows_bwd_saved (the training loop's pair: the forward keeps z, r,
    tanh(.) and r * h, the backward skips its recompute passes) against the fp64 oracle; keep = None: no row list.  The
    forward's output equals impnn_gated_update_rows' bit for bit, and a second backward over the consumed buffer is refused
    at the autograd node (test_wide_state_model_gradients... run through the same pair)."""
    from ionic_mpnn_amd import autograd
    rng = np.random.default_rng(3 * D + rows)
    names = ["Wz", "bz", "Wr", "br", "Wh", "bh", "gamma", "beta"]
    vals = {"Wz": rng.normal(size=(2 * D, D)) / np.sqrt(2 * D), "Wr": rng.normal(size=(2 * D, D)) / np.sqrt(2 * D),
            "Wh": rng.normal(size=(2 * D, D)) / np.sqrt(2 * D), "bz": rng.normal(size=D) * 0.1,
            "br": rng.normal(size=D) * 0.1, "bh": rng.normal(size=D) * 0.1, "gamma": 1 + 0.1 * rng.normal(size=D),
            "beta": 0.1 * rng.normal(size=D)}
    h, agg, go = rng.normal(size=(rows, D)), rng.normal(size=(rows, D)), rng.normal(size=(rows, D))
    sel = (np.arange(rows) if keep is None else np.flatnonzero(rng.random(rows) < keep)).astype(np.int32)
    n = len(sel)
    po = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in vals.items()}
    ho, ao = (torch.tensor(a[sel], dtype=torch.float64, requires_grad=True) for a in (h, agg))
    if n:
        out64 = TR.gated_update(ho, ao, po)
        (out64 * torch.tensor(go[sel])).sum().backward()
    f32 = lambda a: torch.tensor(a, dtype=torch.float32, device=DEV)
    row_list = None
    if keep is not None:
        idx = torch.zeros(rows, dtype=torch.int32, device=DEV)
        idx[:n] = torch.tensor(sel, device=DEV)
        idx[n:] = -7
        row_list = (idx, torch.tensor([n], dtype=torch.int32, device=DEV))
    ts = (f32(h), f32(agg), *[f32(vals[k]) for k in names])
    out, kept = ops.gated_update(*ts, 1e-3, rows=row_list, save=True)
    plain = ops.gated_update(*ts, 1e-3, rows=row_list)
    torch.cuda.synchronize()
    if n:
        assert torch.equal(out[sel], plain[sel])
        close(out[sel], out64.detach(), what="forward output of the listed rows")
    dh, dagg, *dp = autograd._gated_update_backward(ts, 1e-3, f32(go), row_list, kept)[:10]
    torch.cuda.synchronize()
    if n:
        close(dh[sel], ho.grad, what="dh of the listed rows")
        close(dagg[sel], ao.grad, what="dagg of the listed rows")
        for k, g in zip(names, dp):
            close(g, po[k].grad, what=f"d{k}")
    else:
        for g in dp:
            assert float(g.abs().max()) == 0.0
    rest = np.setdiff1d(np.arange(rows), sel)
    if len(rest):
        assert float(dh[rest].abs().max()) == 0.0



def test_embedding_and_pool_backward():
    rng = np.random.default_rng(3)
    B, N, D, V = 6, 11, 16, 9
    ids = rng.integers(0, V, size=(B, N)).astype(np.int32)
    table, go = rng.normal(size=(V, D)), rng.normal(size=(B, D))
    to = torch.tensor(table, dtype=torch.float64, requires_grad=True)
    p = TR.global_sum_pool(torch.nn.functional.embedding(torch.tensor(ids).long(), to), torch.tensor(ids))
    (p * torch.tensor(go)).sum().backward()
    tg = torch.tensor(table, dtype=torch.float32, device=DEV, requires_grad=True)
    ig = torch.tensor(ids, device=DEV)
    pg = ops.global_sum_pool(ops.embed_gather(ig, tg), ig)
    close(pg, p, 1e-6, "pooled")
    (pg * torch.tensor(go, dtype=torch.float32, device=DEV)).sum().backward()
    close(tg.grad, to.grad, what="dtable")


def _tiny_model(kind="viscosity", S=2, seed=5):
    Va, Vb, D, K = 11, 6, 16, 4
    w = weights.init_weights(kind, Va, Vb, atom_dim=D, bond_dim=K, fp_size=12, mixing_size=10, num_steps=S,
                             seed=seed, perturb=True)
    m = MM.build_model(Va, Vb, atom_dim=D, bond_dim=K, fp_size=12, mixing_size=10, num_steps=S, device=DEV)
    m.load_weights(w)
    inp = synthetic.make_batch(24, max_atoms=10, max_edges=16, atom_vocab_size=Va, bond_vocab_size=Vb, min_atoms=3, seed=seed)
    y = np.random.default_rng(seed).normal(1.0, 0.5, size=24).astype(np.float32)
    return m, w, inp, y


def test_whole_model_gradients_match_the_oracle():
    m, w, inp, y = _tiny_model()
    m.compile(train.Adam(1e-3, clipnorm=1.0))
    loss = m._loss(m._to_device(inp), y, training=True)
    loss.backward()
    wo = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    pred = TR.viscosity_forward(wo, inp, torch.float64)
    lo = torch.mean((pred.reshape(-1) - torch.tensor(y, dtype=torch.float64)) ** 2) \
        + 1e-4 * ((wo["cat_fp/kernel"] ** 2).sum() + (wo["an_fp/kernel"] ** 2).sum())
    lo.backward()
    close(loss, lo, 1e-5, "loss")
    for name, t in m.trainable_variables():
        close(t.grad, wo[name].grad, 2e-4, f"grad {name}")


@pytest.mark.parametrize("B,N", [(120, 36), (30, 36)])
def test_wide_state_model_gradients_match_the_oracle(B, N):
    """atom_dim 64 (the wide kernels: matrix-core message backward, 16-row GatedUpdate tiles; at B = 120 the 4 320 atom
    rows per ion switch on the kept-row list, impnn_gated_update_rows[_bwd]) - every parameter gradient of the whole
    viscosity model against fp64 autograd over oracle/torch_ref.py."""
    Va, Vb, D, K, S = 13, 6, 64, 4, 2
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=D, bond_dim=K, fp_size=12, mixing_size=10, num_steps=S, seed=9,
                             perturb=True)
    m = MM.build_model(Va, Vb, atom_dim=D, bond_dim=K, fp_size=12, mixing_size=10, num_steps=S, device=DEV)
    m.load_weights(w)
    inp = synthetic.make_batch(B, max_atoms=N, max_edges=24, atom_vocab_size=Va, bond_vocab_size=Vb, min_atoms=3, seed=9)
    y = np.random.default_rng(9).normal(1.0, 0.5, size=B).astype(np.float32)
    assert (B * N >= MM.TRAIN_ROW_LIST_MIN_ROWS) == (B == 120)
    m.compile(train.Adam(1e-3, clipnorm=1.0))
    loss = m._loss(m._to_device(inp), y, training=True)
    loss.backward()
    wo = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    pred = TR.viscosity_forward(wo, inp, torch.float64)
    lo = torch.mean((pred.reshape(-1) - torch.tensor(y, dtype=torch.float64)) ** 2) \
        + 1e-4 * ((wo["cat_fp/kernel"] ** 2).sum() + (wo["an_fp/kernel"] ** 2).sum())
    lo.backward()
    close(loss, lo, 1e-5, "loss")
    for name, t in m.trainable_variables():
        close(t.grad, wo[name].grad, 2e-4, f"grad {name}")


@pytest.mark.parametrize("D,B", [(64, 6), (32, 5)])
def test_gradients_at_the_explicit_hydrogen_padded_shape(D, B):
    """Training on the padded shape of the reference's real data sets (N = 160, E = 640: src/featurize.py:45,
    train_viscosity.py:95,288-289) - long edge lists per molecule in the message / Reduce adjoints, kept activations in
    the GatedUpdate pair: every parameter gradient against fp64 autograd over oracle/torch_ref.py."""
    Va, Vb, K, S, N, E = 13, 6, 4, 2, 160, 640
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=D, bond_dim=K, fp_size=12, mixing_size=10, num_steps=S, seed=19,
                             perturb=True)
    m = MM.build_model(Va, Vb, atom_dim=D, bond_dim=K, fp_size=12, mixing_size=10, num_steps=S, device=DEV)
    m.load_weights(w)
    inp = synthetic.make_explicit_h_batch(B, max_atoms=N, max_edges=E, atom_vocab_size=Va, bond_vocab_size=Vb, seed=19)
    y = np.random.default_rng(19).normal(1.0, 0.5, size=B).astype(np.float32)
    m.compile(train.Adam(1e-3, clipnorm=1.0))
    loss = m._loss(m._to_device(inp), y, training=True)
    loss.backward()
    wo = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    pred = TR.viscosity_forward(wo, inp, torch.float64)
    lo = torch.mean((pred.reshape(-1) - torch.tensor(y, dtype=torch.float64)) ** 2) \
        + 1e-4 * ((wo["cat_fp/kernel"] ** 2).sum() + (wo["an_fp/kernel"] ** 2).sum())
    lo.backward()
    close(loss, lo, 1e-5, "loss")
    for name, t in m.trainable_variables():
        close(t.grad, wo[name].grad, 2e-4, f"grad {name}")


@pytest.mark.parametrize("D", [16, 64])
def test_gradients_with_single_atom_anions_and_few_bond_types(D):
    """The statistics real ionic-liquid data has - halide-like anions (one atom, no bond) and two bond types - through
    the whole training graph: every parameter gradient against fp64 autograd over oracle/torch_ref.py."""
    Va, Vb, K, S, B = 13, 6, 4, 2, 120
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=D, bond_dim=K, fp_size=12, mixing_size=10, num_steps=S, seed=19,
                             perturb=True)
    m = MM.build_model(Va, Vb, atom_dim=D, bond_dim=K, fp_size=12, mixing_size=10, num_steps=S, device=DEV)
    m.load_weights(w)
    inp = synthetic.make_batch(B, max_atoms=36, max_edges=72, atom_vocab_size=Va, bond_vocab_size=Vb, min_atoms=3, seed=19)
    for p_ in ("cat", "an"):
        inp[p_ + "_bond"] = np.where(inp[p_ + "_bond"] > 0, 1 + (inp[p_ + "_bond"] & 1), 0).astype(np.int32)
    inp["an_atom"][:, 1:] = 0
    inp["an_bond"][:] = 0
    inp["an_connectivity"][:] = 0
    y = np.random.default_rng(19).normal(1.0, 0.5, size=B).astype(np.float32)
    m.compile(train.Adam(1e-3, clipnorm=1.0))
    loss = m._loss(m._to_device(inp), y, training=True)
    loss.backward()
    wo = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    pred = TR.viscosity_forward(wo, inp, torch.float64)
    lo = torch.mean((pred.reshape(-1) - torch.tensor(y, dtype=torch.float64)) ** 2) \
        + 1e-4 * ((wo["cat_fp/kernel"] ** 2).sum() + (wo["an_fp/kernel"] ** 2).sum())
    lo.backward()
    close(loss, lo, 1e-5, "loss")
    for name, t in m.trainable_variables():
        close(t.grad, wo[name].grad, 2e-4, f"grad {name}")


def test_melting_point_model_gradients_match_the_oracle():
    """train_melting_point.py: bond_dim = atom_dim^2 (the per-bond-type schedule is mandatory), l2(1e-5) on the
    fingerprint and hidden Dense kernels."""
    Va, Vb, D = 9, 5, 8
    w = weights.init_weights("melting_point", Va, Vb, atom_dim=D, bond_dim=D * D, fp_size=12, mixing_size=10,
                             num_steps=2, seed=4, perturb=True)
    m = MM.build_melting_point_model(Va, Vb, atom_dim=D, fp_size=12, mixing_size=10, num_steps=2, device=DEV)
    m.load_weights(w)
    inp = synthetic.make_batch(12, max_atoms=9, max_edges=14, atom_vocab_size=Va, bond_vocab_size=Vb, min_atoms=3,
                               seed=4, with_temperature=False)
    y = np.random.default_rng(4).normal(0.0, 1.0, size=12).astype(np.float32)
    m.compile()
    loss = m._loss(m._to_device(inp), y, training=True)
    loss.backward()
    wo = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    pred = TR.melting_point_forward(wo, inp, torch.float64)
    lo = torch.mean((pred.reshape(-1) - torch.tensor(y, dtype=torch.float64)) ** 2) + 1e-5 * (
        (wo["cat_fp/kernel"] ** 2).sum() + (wo["an_fp/kernel"] ** 2).sum() + (wo["mp_hidden/kernel"] ** 2).sum())
    lo.backward()
    close(loss, lo, 1e-5, "loss")
    for name, t in m.trainable_variables():
        close(t.grad, wo[name].grad, 2e-4, f"grad {name}")


@pytest.mark.parametrize("kind", ["viscosity", "melting_point"])
@pytest.mark.parametrize("sinks", [False, True])
def test_head_node_matches_the_layer_by_layer_head(kind, sinks):
    """impnn_model_head_tensors / impnn_model_head_bwd against the torch-op head (Dense, relu, Add, softplus, clip,
    LogViscosity), including samples inside the clip plateaus (zero gradient through B or C) and a batch that is not
    a multiple of the 8 samples a workgroup takes."""
    Va, Vb, D, K, B = 11, 6, 16, 4, 203
    build = MM.build_model if kind == "viscosity" else MM.build_melting_point_model
    kw = dict(atom_dim=D, fp_size=12, mixing_size=10, num_steps=1, device=DEV)
    if kind == "viscosity":
        kw["bond_dim"] = K
    m = build(Va, Vb, **kw)
    m.load_weights(weights.init_weights(kind, Va, Vb, atom_dim=D, bond_dim=K if kind == "viscosity" else D * D,
                                        fp_size=12, mixing_size=10, num_steps=1, seed=9, perturb=True))
    g = torch.Generator(device="cpu").manual_seed(3)
    pc = (torch.randn(B, D, generator=g) * 3).to(DEV).requires_grad_(True)
    pa = (torch.randn(B, D, generator=g) * 3).to(DEV).requires_grad_(True)
    T = (torch.rand(B, 1, generator=g) * 150 + 250).to(DEV)
    go = torch.randn(B, 1, generator=g).to(DEV)
    if kind == "viscosity":
        with torch.no_grad():  # push b and c of some samples onto the clip plateaus
            m.visc_params.kernel.mul_(6.0)
    params = m._head_tensors()
    for t in params:
        t.requires_grad_(True)
        t.grad = None
    ref = m.head(pc, pa, T, trace={}, differentiable=True)
    (ref * go).sum().backward()
    want = [t.grad.clone() for t in params] + [pc.grad.clone(), pa.grad.clone()]
    if kind == "viscosity":
        vp = m.visc_params(m.mix([m.cat_proj(m.branches["cat"]["fp"](pc)), m.an_proj(m.branches["an"]["fp"](pa))]))
        sp = torch.nn.functional.softplus(vp.detach())
        assert (sp[:, 1] > 20).any() and (sp[:, 2] < 0.1).any() and ((sp[:, 1] < 20) & (sp[:, 2] > 0.1)).any()
    pc.grad = pa.grad = None
    for t in params:
        t.grad = torch.zeros_like(t) if sinks else None
    got = m.head(pc, pa, T, differentiable=True)
    close(got, ref, 1e-5, "head forward")
    (got * go).sum().backward()
    have = [t.grad for t in params] + [pc.grad, pa.grad]
    for i, (a, b) in enumerate(zip(have, want)):
        close(a, b, 1e-4, f"head gradient {i}")


@pytest.mark.parametrize("kind,D", [("viscosity", 16), ("melting_point", 16), ("viscosity", 128)])
def test_loss_node_matches_mse_plus_penalties(kind, D):
    """impnn_model_head_loss[_bwd] (head + keras mse + l2 penalties in one launch each) against the torch composition
    of the layer-by-layer head, train.mse and regularization_loss(); the loss gradient arrives as a device scalar
    (0.37 here, as on a data-parallel rank); repeated calls reuse the workspace (arrival counter back at zero)."""
    from ionic_mpnn_amd import autograd
    Va, Vb, K, B = 11, 6, 4, 203
    build = MM.build_model if kind == "viscosity" else MM.build_melting_point_model
    kw = dict(atom_dim=D, fp_size=12, mixing_size=10, num_steps=1, device=DEV)
    if kind == "viscosity":
        kw["bond_dim"] = K
    m = build(Va, Vb, **kw)
    m.load_weights(weights.init_weights(kind, Va, Vb, atom_dim=D, bond_dim=K if kind == "viscosity" else D * D,
                                        fp_size=12, mixing_size=10, num_steps=1, seed=9, perturb=True))
    m.fp_l2 = 0.03  # large enough to matter next to the data term
    g = torch.Generator(device="cpu").manual_seed(3)
    pc = (torch.randn(B, D, generator=g) * 2).to(DEV).requires_grad_(True)
    pa = (torch.randn(B, D, generator=g) * 2).to(DEV).requires_grad_(True)
    T = (torch.rand(B, 1, generator=g) * 150 + 250).to(DEV)
    y = torch.randn(B, 1, generator=g).to(DEV)
    params = m._head_tensors()
    for t in params:
        t.requires_grad_(True)
        t.grad = None
    ref = train.mse(y, m.head(pc, pa, T, trace={}, differentiable=True)) + m.regularization_loss()
    (ref * 0.37).backward()
    want = [t.grad.clone() for t in params] + [pc.grad.clone(), pa.grad.clone()]
    pc.grad = pa.grad = None
    for t in params:
        t.grad = None
    ws = torch.zeros(1024, device=DEV)
    k = {"viscosity": 0, "melting_point": 1}[kind]
    losses = []
    for rep in range(3):
        loss = autograd.ModelHeadLoss.apply(k, 12, 10, m._head_l2(), ws, pc, pa, T if k == 0 else None, y, *params)
        losses.append(loss.detach().clone())
    assert torch.equal(losses[0], losses[1]) and torch.equal(losses[1], losses[2])
    close(loss, ref, 1e-5, "loss")
    (loss * 0.37).backward()
    have = [t.grad for t in params] + [pc.grad, pa.grad]
    for i, (a, b) in enumerate(zip(have, want)):
        close(a, b, 1e-4, f"loss-node gradient {i}")


def test_config5_shape_trains_without_nonfinite_gradients():
    """D=128, S=6 (SURVEY config 5): untrained pre-activations of the viscosity head exceed 88, where a naive
    log1p(exp(x)) differentiates to NaN."""
    Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
    m = MM.build_model(Va, Vb, atom_dim=128, bond_dim=8, num_steps=6, device=DEV)
    m.load_weights(weights.init_weights("viscosity", Va, Vb, atom_dim=128, bond_dim=8, num_steps=6, seed=1))
    inp = synthetic.make_batch(48, seed=0)
    y = np.random.default_rng(0).normal(4.0, 1.0, size=48).astype(np.float32)
    m.compile(train.Adam(1e-3, clipnorm=1.0))
    losses = [float(m.train_on_batch(m._to_device(inp), y)) for _ in range(8)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert all(bool(torch.isfinite(t).all()) for _, t in m.trainable_variables())


def test_adam_clipnorm_step_matches_the_oracle():
    rng = np.random.default_rng(0)
    shapes = [(7, 5), (33,), (4, 8, 8), (1,)]
    ws = [rng.normal(size=s).astype(np.float32) for s in shapes]
    vars_ = [torch.tensor(a, device=DEV, requires_grad=True) for a in ws]
    opt = train.Adam(1e-3, clipnorm=1.0)
    opt.build(vars_)
    ref = [(a.astype(np.float64), np.zeros(a.shape), np.zeros(a.shape)) for a in ws]
    for t in range(1, 6):
        gs = [(rng.normal(size=s) * (3.0 if i % 2 else 0.05)).astype(np.float32) for i, s in enumerate(shapes)]
        for v, g in zip(vars_, gs):
            v.grad.copy_(torch.tensor(g, device=DEV))
        opt.apply_gradients()
        opt.zero_grad()
        ref = [TO.adam_step(w_, g, m_, v_, t, clipnorm=1.0) for (w_, m_, v_), g in zip(ref, gs)]
        for v, (w_, _, _) in zip(vars_, ref):
            close(v, w_, 2e-6, f"step {t}")


def test_fit_reduces_the_loss_and_early_stopping_restores_the_best_weights():
    m, _, inp, y = _tiny_model(S=1, seed=7)
    m.compile(train.Adam(1e-2, clipnorm=1.0))
    first = m.evaluate(m._to_device(inp), y)
    es = train.EarlyStopping(monitor="val_loss", patience=3, restore_best_weights=True)
    hist = m.fit(inp, y, validation_data=(inp, y), epochs=25, batch_size=8, callbacks=[es], seed=0)
    assert hist.history["loss"][-1] < 0.6 * first
    assert len(hist.history["val_loss"]) == len(hist.history["loss"])
    best = min(hist.history["val_loss"])
    assert abs(m.evaluate(m._to_device(inp), y) - best) <= 1e-5 * max(best, 1.0)   # restored
    # inference after training runs the fused encoder on the updated weights
    a = m.predict(inp)
    b = m.predict(inp, fused=False)
    close(a, b, 1e-5, "fused vs layered after training")


def test_gather_rows_and_resident_fit_follow_the_eager_fit():
    """impnn_gather_rows is bit-exact index_select for the 8 batch tensors in one launch; fit(graph=True) - the gather
    as first node of the captured step, only the row indices refreshed per replay - walks the trajectory of the
    eager fit (same shuffles)."""
    rng = np.random.default_rng(2)
    srcs = [torch.tensor(rng.integers(0, 99, size=(57, 10)), dtype=torch.int32, device=DEV),
            torch.tensor(rng.integers(0, 99, size=(57, 16, 2)), dtype=torch.int32, device=DEV),
            torch.tensor(rng.normal(size=(57, 1)), dtype=torch.float32, device=DEV)]
    rows = torch.tensor(rng.integers(0, 57, size=23), dtype=torch.int64, device=DEV)
    dsts = [torch.empty((23,) + tuple(t.shape[1:]), dtype=t.dtype, device=DEV) for t in srcs]
    ops.gather_rows(srcs, dsts, rows)
    for sx, dx in zip(srcs, dsts):
        assert torch.equal(dx, sx[rows])
    hist = {}
    for graph in (False, True):
        m, w, inp, y = _tiny_model()
        big = synthetic.make_batch(72, max_atoms=10, max_edges=16, atom_vocab_size=11, bond_vocab_size=6, min_atoms=3,
                                   seed=8)
        yb = np.random.default_rng(8).normal(1.0, 0.5, size=72).astype(np.float32)
        m.compile(train.Adam(1e-3, clipnorm=1.0))
        hist[graph] = m.fit(big, yb, epochs=3, batch_size=24, seed=5, graph=graph).history["loss"]
    np.testing.assert_allclose(hist[True], hist[False], rtol=2e-4)


def test_melting_point_fit_graphed_follows_eager():
    """bond_dim = atom_dim^2 keeps the per-layer type-matrix nodes (GEMM-shaped) whose gradients reach the leaves
    through AccumulateGrad from the side stream of the anion chain: the captured step must still follow the eager one."""
    import warnings
    Va, Vb, D = 9, 5, 8
    w = weights.init_weights("melting_point", Va, Vb, atom_dim=D, bond_dim=D * D, fp_size=12, mixing_size=10,
                             num_steps=2, seed=4, perturb=True)
    x = synthetic.make_batch(96, max_atoms=9, max_edges=14, atom_vocab_size=Va, bond_vocab_size=Vb, min_atoms=3, seed=4,
                             with_temperature=False)
    y = np.random.default_rng(4).normal(0.0, 1.0, size=96).astype(np.float32)
    hist = {}
    for graph in (False, True):
        m = MM.build_melting_point_model(Va, Vb, atom_dim=D, fp_size=12, mixing_size=10, num_steps=2, device=DEV)
        m.load_weights(w)
        m.compile(train.Adam(1e-3, clipnorm=1.0))
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # torch notes the (intended) stream of the side chain's AccumulateGrad
            hist[graph] = m.fit(x, y, epochs=3, batch_size=32, seed=0, graph=graph).history["loss"]
    np.testing.assert_allclose(hist[True], hist[False], rtol=2e-4)


def test_graphed_train_step_follows_the_eager_trajectory():
    """The captured hipGraph of (forward, backward, Adam) must walk the same path as eager steps: same losses and
    same weights after several different mini-batches (float atomics in the embedding / message backward allow
    last-bit differences only)."""
    ma, w, inp, y = _tiny_model(S=2, seed=11)
    mb, _, _, _ = _tiny_model(S=2, seed=11)
    for m in (ma, mb):
        m.compile(train.Adam(1e-3, clipnorm=1.0))
    d = ma._to_device(inp)
    batches = [np.arange(0, 8), np.arange(8, 16), np.arange(16, 24), np.arange(4, 12)]
    pick = lambda idx: ({k: v[torch.from_numpy(idx).to(DEV)] for k, v in d.items()}, y[idx])
    g = train.GraphedTrainStep(mb, *pick(batches[0]))
    for _, (ta, tb) in zip(range(99), zip(ma.trainable_variables(), mb.trainable_variables())):
        assert torch.equal(ta[1], tb[1])                       # capture left the weights untouched
    assert mb.optimizer.iterations == 0
    for idx in batches:
        la = ma.train_on_batch(*pick(idx))
        lb = g(*pick(idx)).clone()
        close(lb, la, 1e-5, "loss")
    assert ma.optimizer.iterations == mb.optimizer.iterations == len(batches)
    for (n, ta), (_, tb) in zip(ma.trainable_variables(), mb.trainable_variables()):
        close(tb, ta, 1e-4, f"weights {n}")
    # fit() uses the graph for full batches and eager steps for the remainder
    h1 = ma.fit(inp, y, epochs=2, batch_size=10, seed=3, graph=False)
    h2 = mb.fit(inp, y, epochs=2, batch_size=10, seed=3, graph=True)
    close(np.array(h2.history["loss"]), np.array(h1.history["loss"]), 1e-4, "fit loss, graph vs eager")


def test_graph_replay_survives_allocator_churn():
    """Every pointer a captured step reads or writes (static inputs, gradient sinks, per-node workspaces, the loss
    workspace, the Adam state) must stay valid for the graph's whole life: between replays this test frees and
    re-allocates hundreds of MB through torch's caching allocator, scribbles NaNs over what it gets, and drops the
    Python references the capture left behind (gc).  The trajectory must still equal the eager one.
    ("hipMemsetAsync in a captured graph": the lifetime audit of the captured buffers.)"""
    import gc
    ma, w, inp, y = _tiny_model(S=2, seed=13)
    mb, _, _, _ = _tiny_model(S=2, seed=13)
    for m in (ma, mb):
        m.compile(train.Adam(1e-3, clipnorm=1.0))
    d = ma._to_device(inp)
    pick = lambda idx: ({k: v[torch.from_numpy(idx).to(DEV)] for k, v in d.items()}, y[idx])
    g = train.GraphedTrainStep(mb, *pick(np.arange(0, 8)))
    rng = np.random.default_rng(0)
    for it in range(6):
        junk = [torch.full((int(rng.integers(1, 64)) << 18,), float("nan"), device=DEV) for _ in range(6)]
        del junk
        gc.collect()
        torch.cuda.empty_cache()                      # hand the eager pool's blocks back; the graph's pool must stay
        junk = [torch.full((1 << 22,), float("nan"), device=DEV) for _ in range(4)]
        idx = rng.permutation(24)[:8]
        la = ma.train_on_batch(*pick(idx))
        lb = g(*pick(idx)).clone()
        close(lb, la, 1e-5, f"loss at replay {it}")
        del junk
    for (n, ta), (_, tb) in zip(ma.trainable_variables(), mb.trainable_variables()):
        assert torch.isfinite(tb).all(), n
        close(tb, ta, 2e-4, f"weights {n}")


def test_weight_file_round_trip(tmp_path):
    """f3: config + variables under their Keras-style names survive save_weights -> from_config + load_weights."""
    from ionic_mpnn_amd import layers as LL
    LL.reset_uids()  # keras auto-names (gated_update_3, ...) count from the start of a session
    m, _, inp, _ = _tiny_model(S=2, seed=9)
    path = tmp_path / "viscosity_final.npz"
    m.save_weights(path)
    cfg, w = MM.MPNNModel.load_weight_file(path)
    assert cfg["atom_dim"] == 16 and cfg["num_steps"] == 2 and "cat_gu_1/dense_z/kernel" in w
    m2 = MM.MPNNModel.from_config(cfg, device=DEV)
    m2.load_weights(path)
    m3 = MM.load_model(path, custom_objects={"BondMatrixMessage": None}, device=DEV)   # keras.models.load_model analogue
    assert np.array_equal(m.predict(inp), m3.predict(inp))
    assert [l.name for l in m2.layers] == [l.name for l in m.layers]
    assert np.array_equal(m.predict(inp), m2.predict(inp))
    lc = m.get_config()["layers"]
    assert {"atom_dim": 16, "bond_dim": 4}.items() <= next(c["config"] for c in lc if c["class_name"] == "BondMatrixMessage").items()
    # the reference's own file name (train_viscosity.py:354 -> train_melting_point_transfer.py:78): the name is kept
    # verbatim (numpy would append ".npz" to a plain path)
    kpath = tmp_path / "viscosity_final.keras"
    m.save(str(kpath))
    assert kpath.exists() and not (tmp_path / "viscosity_final.keras.npz").exists()
    m4 = MM.load_model(str(kpath), device=DEV)
    assert np.array_equal(m.predict(inp), m4.predict(inp))
    m2.load_weights(str(kpath))


def test_trainer_flow_end_to_end(monkeypatch):
    """records -> resident dataset (GPU batch assembly) -> fit (graphed steps) -> predict: the flow of
    train_viscosity.py main() (tools/example_train_viscosity.py)."""
    import importlib.util
    import sys
    from pathlib import Path
    path = Path(__file__).resolve().parents[1] / "tools" / "example_train_viscosity.py"
    spec = importlib.util.spec_from_file_location("example_train_viscosity", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    monkeypatch.setattr(sys, "argv", ["example", "--records", "160", "--epochs", "6"])
    out = mod.main()
    assert out["epochs_run"] == 6 and np.isfinite(out["last_loss"]) and out["last_loss"] < out["first_loss"]
    assert all(np.isfinite(out[k]) for k in ("train_r2", "dev_mae", "test_mae"))


def _fit_rank(rank, world, port, out_dir):
    """One data-parallel rank of model.fit: ranks share cuda:0 (one-GPU box), gloo carries the collectives."""
    import os
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
    os.environ.update({"RANK": str(rank), "LOCAL_RANK": "0", "WORLD_SIZE": str(world), "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    from ionic_mpnn_amd import dist as idist
    idist.init_distributed(backend="gloo")
    m, _, inp, y = _tiny_model(S=1, seed=7)
    m.compile(train.Adam(1e-2, clipnorm=1.0))
    # only rank 0 names a seed: the others must end up with its permutation stream (broadcast), and 23 samples at
    # batch 10 leave a last mini-batch of 3 that is cut 1 + 2 - every rank still runs the same number of steps
    sub = {k: v[:23] for k, v in inp.items()}
    hist = m.fit(sub, y[:23], epochs=3, batch_size=10, seed=11 if rank == 0 else None)
    np.savez(Path(out_dir) / f"fit_{rank}.npz", loss=np.array(hist.history["loss"]), **m.state_dict())
    torch.distributed.destroy_process_group()


def test_fit_under_torch_distributed_two_ranks(tmp_path):
    """fit() under torch.distributed - shared shuffle, sharded mini-batches, same step count on every rank.
    Both ranks must finish with identical weights and the same logged loss, equal (to fp32 reassociation) to the
    single-process run with rank 0's seed."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_fit_rank, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    for r, p in enumerate(procs):
        if p.is_alive():
            p.kill()
            pytest.fail(f"rank {r} hung")
        assert p.exitcode == 0, f"rank {r} exit code {p.exitcode}"
    a, b = np.load(tmp_path / "fit_0.npz"), np.load(tmp_path / "fit_1.npz")
    for k in a.files:
        assert np.array_equal(a[k], b[k]), f"ranks disagree on {k}"
    m, _, inp, y = _tiny_model(S=1, seed=7)
    m.compile(train.Adam(1e-2, clipnorm=1.0))
    hist = m.fit({k: v[:23] for k, v in inp.items()}, y[:23], epochs=3, batch_size=10, seed=11, graph=False)
    close(a["loss"], np.array(hist.history["loss"]), 1e-4, "distributed vs single-process loss history")
    sd = m.state_dict()
    for k in ("cat_gu_0/dense_z/kernel", "atom_embedding", "visc_params/kernel"):
        close(a[k], sd[k], 2e-3, f"distributed vs single-process {k}")
