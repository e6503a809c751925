"""GPU parity of the fused encoder (impnn_encoder_fused: the graphs/sec path) against the oracle
on golden fixtures and random shapes, against the layer-at-a-time HIP path, and - at
BASELINE.json's full batch 4096 - through size-independent properties."""
import numpy as np
import pytest
import torch

from conftest import assert_close, load_case
from ionic_mpnn_amd import model as MM
from ionic_mpnn_amd import ops, synthetic, weights
from oracle import mpnn_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def to_dev(inputs):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(DEV) for k, v in inputs.items()}


def make_model(w, Va, Vb, D=32, K=8, fp=32, mix=20, mode="auto"):
    m = MM.build_model(Va, Vb, atom_dim=D, bond_dim=K, fp_size=fp, mixing_size=mix, num_steps=weights.num_steps_of(w),
                       device=DEV)
    m.load_weights(w)
    m.encoder_mode = mode
    return m


MODES = ["f32t", "f32x3", "f32", "f16x2"]


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("name", ["config2_b8", "config2_perturbed_b6"])
def test_fused_encoder_vs_golden(name, mode):
    _, inp, w, outs = load_case(name)
    m = make_model(w, *w["atom_embedding"].shape[:1], w["bond_embedding"].shape[0], mode=mode)
    pc, pa = m.encode_pooled(to_dev(inp), fused=True)
    assert_close(pc.cpu().numpy(), outs["cat/pooled"], what="cat pooled")
    assert_close(pa.cpu().numpy(), outs["an/pooled"], what="an pooled")
    y = m(inp, fused=True).cpu().numpy()
    assert_close(y, outs["final"], what="log_eta")
    # layered HIP path: same answer, and per-layer tensors match the oracle's
    tr = {}
    yl = m(inp, fused=False, trace=tr).cpu().numpy()
    assert_close(yl, outs["final"], what="log_eta layered")
    for k in ("cat/m0", "cat/agg1", "an/h2", "an/pooled", "cat/fp", "mixed"):
        assert_close(tr[k].cpu().numpy(), outs[k], what=k)


@pytest.mark.parametrize("N,E,K,S,B,seed", [(40, 80, 8, 3, 64, 1), (40, 80, 8, 4, 257, 2), (12, 20, 4, 2, 100, 3),
                                            (7, 30, 1, 1, 50, 4), (64, 120, 8, 2, 40, 5), (40, 80, 5, 0, 30, 6),
                                            (1, 0, 8, 2, 9, 7), (100, 240, 8, 1, 16, 8), (128, 512, 3, 1, 5, 9)])
@pytest.mark.parametrize("mode", MODES)
def test_fused_encoder_random_shapes(N, E, K, S, B, seed, mode):
    Va, Vb = 30, 11
    inp = synthetic.make_batch(B, max_atoms=N, max_edges=E, atom_vocab_size=Va, bond_vocab_size=Vb,
                               min_atoms=min(3, N), seed=seed)
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=32, bond_dim=K, num_steps=S, seed=seed + 100, perturb=True)
    m = make_model(w, Va, Vb, K=K, mode=mode)
    assert m.fused_supported(N, E)
    pc, pa = m.encode_pooled(to_dev(inp), fused=True)
    rc = O.encode(w, "cat", inp["cat_atom"], inp["cat_bond"], inp["cat_connectivity"], pooled_only=True)
    ra = O.encode(w, "an", inp["an_atom"], inp["an_bond"], inp["an_connectivity"], pooled_only=True)
    assert_close(pc.cpu().numpy(), rc, what="cat pooled")
    assert_close(pa.cpu().numpy(), ra, what="an pooled")


@pytest.mark.parametrize("mode", MODES)
def test_fused_encoder_adversarial_graphs(mode):
    """Edges that name padding atoms, self loops, 4x duplicated bonds (trainer expansion), id-0 holes,
    all-padding molecules, dense in-degree - the general contract, not just tree graphs."""
    rng = np.random.default_rng(42)
    B, N, E, Va, Vb, K, S = 48, 24, 64, 20, 7, 8, 3
    ids = rng.integers(0, Va, size=(B, N)).astype(np.int32)          # zeros anywhere (holes)
    ids[0] = 0                                                        # an all-padding molecule
    conn = rng.integers(0, N, size=(B, E, 2)).astype(np.int32)        # any atom, incl. padding atoms and 0
    conn[1] = 5                                                       # 64 self loops on atom 5
    conn[2, :, 1] = 3                                                 # in-degree 64 on one atom
    conn[:, 48:] = 0
    bond = rng.integers(0, Vb, size=(B, E)).astype(np.int32)
    # trainer-style 4x duplication for some molecules
    e4, b4 = O.preprocess_edges_and_bonds([[(0, 1), (1, 0), (1, 2), (2, 1), (2, 3), (3, 2)]] * 4,
                                          [[1, 1, 2, 2, 3, 3]] * 4, E // 2)
    conn[3:7], bond[3:7] = e4, b4
    inp = {"cat_atom": ids, "cat_bond": bond, "cat_connectivity": conn,
           "an_atom": ids[::-1].copy(), "an_bond": bond[::-1].copy(), "an_connectivity": conn[::-1].copy()}
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=32, bond_dim=K, num_steps=S, seed=9, perturb=True)
    m = make_model(w, Va, Vb, K=K, mode=mode)
    pc, pa = m.encode_pooled(to_dev(inp), fused=True)
    rc = O.encode(w, "cat", ids, bond, conn, pooled_only=True)
    ra = O.encode(w, "an", inp["an_atom"], inp["an_bond"], inp["an_connectivity"], pooled_only=True)
    assert_close(pc.cpu().numpy(), rc, what="cat pooled")
    assert_close(pa.cpu().numpy(), ra, what="an pooled")
    assert float(pc[0].abs().max()) == 0.0


@pytest.mark.parametrize("mode", MODES)
def test_fused_encoder_in_degrees_around_the_plan_ticket_list(mode):
    """The typed plan ranks a row's in-edges through a 16-entry ticket list and takes a slower path beyond it:
    in-degrees 15, 16, 17, 18 and 40 on atoms of one molecule, edge slots interleaved (the summation order is the
    edge-slot order, models/layers.py:78-82), beside ordinary molecules."""
    rng = np.random.default_rng(77)
    B, N, E, Va, Vb, K, S = 24, 30, 120, 20, 7, 8, 3
    ids = rng.integers(1, Va, size=(B, N)).astype(np.int32)
    conn = np.zeros((B, E, 2), dtype=np.int32)
    bond = np.zeros((B, E), dtype=np.int32)
    for b in range(B):
        tgt = np.concatenate([np.full(15, 3), np.full(16, 4), np.full(17, 5), np.full(18, 6), np.full(40, 7)])
        tgt = np.concatenate([tgt, rng.integers(8, N, size=E - tgt.size - 6)])
        rng.shuffle(tgt)                                              # interleave the slots of the five hubs
        n = tgt.size
        conn[b, :n, 1] = tgt
        conn[b, :n, 0] = rng.integers(1, N, size=n)
        bond[b, :n] = rng.integers(1, Vb, size=n)
    inp = {"cat_atom": ids, "cat_bond": bond, "cat_connectivity": conn,
           "an_atom": ids[::-1].copy(), "an_bond": bond[::-1].copy(), "an_connectivity": conn[::-1].copy()}
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=32, bond_dim=K, num_steps=S, seed=5, perturb=True)
    m = make_model(w, Va, Vb, K=K, mode=mode)
    pc, pa = m.encode_pooled(to_dev(inp), fused=True)
    rc = O.encode(w, "cat", ids, bond, conn, pooled_only=True)
    ra = O.encode(w, "an", inp["an_atom"], inp["an_bond"], inp["an_connectivity"], pooled_only=True)
    assert_close(pc.cpu().numpy(), rc, what="cat pooled")
    assert_close(pa.cpu().numpy(), ra, what="an pooled")


@pytest.mark.parametrize("B", [4096, 20000])
@pytest.mark.parametrize("mode", ["f32t", "f32x3", "f32"])
def test_fused_encoder_single_atom_anions(mode, B):
    """Halide-like anions (one atom, no bond) against ordinary cations: the ions' row counts differ 25-fold, so the
    anion's few persistent workgroups take shares of several thousand molecules each and chunks of 256 molecules."""
    Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
    inp = synthetic.make_batch(B, seed=5)
    inp["an_atom"][:, 1:] = 0
    inp["an_bond"][:] = 0
    inp["an_connectivity"][:] = 0
    w = weights.init_weights("viscosity", Va, Vb, num_steps=3, seed=6, perturb=True)
    idx = np.concatenate([np.arange(8), np.random.default_rng(1).choice(B, size=16, replace=False), np.arange(B - 8, B)])
    ra = O.encode(w, "an", inp["an_atom"][idx], inp["an_bond"][idx], inp["an_connectivity"][idx], pooled_only=True)
    rc = O.encode(w, "cat", inp["cat_atom"][idx], inp["cat_bond"][idx], inp["cat_connectivity"][idx], pooled_only=True)
    m = make_model(w, Va, Vb, mode=mode)
    pc, pa = m.encode_pooled(to_dev(inp), fused=True)
    assert bool(torch.isfinite(pc).all()) and bool(torch.isfinite(pa).all())
    assert_close(pa.cpu().numpy()[idx], ra, what="an pooled")
    assert_close(pc.cpu().numpy()[idx], rc, what="cat pooled")
    # every anion is the same computation on its one embedding row: equal atom ids, equal rows - everywhere in the batch
    ids = inp["an_atom"][:, 0]
    first = {int(v): int(np.argmax(ids == v)) for v in np.unique(ids)}
    ref_rows = pa[torch.as_tensor([first[int(v)] for v in ids], device=pa.device)]
    assert torch.equal(pa, ref_rows)


@pytest.mark.parametrize("mode", ["f32t", "f32x3"])
def test_fused_encoder_explicit_hydrogen_cations_with_single_atom_anions(mode):
    """The real data sets' padded shape (N = 160, E = 640) with few bond types and halide-like anions, batch 2048."""
    Va, Vb, B = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, 2048
    inp = synthetic.make_explicit_h_batch(B, seed=25, bond_type_probs=(0.6, 0.25, 0.1, 0.05))
    inp["an_atom"][:, 1:] = 0
    inp["an_bond"][:] = 0
    inp["an_connectivity"][:] = 0
    w = weights.init_weights("viscosity", Va, Vb, num_steps=3, seed=26, perturb=True)
    idx = np.concatenate([np.arange(3), np.random.default_rng(3).choice(B, size=6, replace=False), np.arange(B - 3, B)])
    ra = O.encode(w, "an", inp["an_atom"][idx], inp["an_bond"][idx], inp["an_connectivity"][idx], pooled_only=True)
    rc = O.encode(w, "cat", inp["cat_atom"][idx], inp["cat_bond"][idx], inp["cat_connectivity"][idx], pooled_only=True)
    m = make_model(w, Va, Vb, mode=mode)
    pc, pa = m.encode_pooled(to_dev(inp), fused=True)
    assert int(getattr(m, "overflow_fallbacks", 0)) == 0
    assert_close(pa.cpu().numpy()[idx], ra, what="an pooled")
    assert_close(pc.cpu().numpy()[idx], rc, what="cat pooled")


def test_unsupported_shapes_fall_back_to_layered_hip():
    # melting-point model: K = D*D (train_melting_point.py:146) is outside the fused kernel
    _, inp, w, outs = load_case("tiny_melting_point")
    m = MM.build_melting_point_model(15, 7, atom_dim=8, fp_size=8, mixing_size=6, num_steps=2, device=DEV)
    m.load_weights(w)
    assert not m.fused_supported(10, 20)
    with pytest.raises(ops.EncoderUnsupported):
        m.encode_pooled(to_dev(inp), fused=True)
    tr = {}
    y = m(inp, trace=tr).cpu().numpy()                                # auto -> layered, typed schedule
    assert_close(y, outs["final"], what="mp_out")
    assert_close(tr["cat/pooled"].cpu().numpy(), outs["cat/pooled"], what="cat pooled")
    # dense bond_state through the literal layer signature agrees too
    pd = m.encode_layered("an", *[to_dev(inp)[k] for k in ("an_atom", "an_bond", "an_connectivity")], typed=False)
    assert_close(pd.cpu().numpy(), outs["an/pooled"], what="an pooled dense")


# ------------------------------------------------------------------ full size (B=4096): properties
@pytest.fixture(scope="module", params=MODES)
def full(request):
    inp = synthetic.make_batch(4096, seed=0)
    w = weights.init_weights("viscosity", synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, num_steps=3, seed=1)
    m = make_model(w, synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, mode=request.param)
    d = to_dev(inp)
    pc, pa = m.encode_pooled(d, fused=True)
    torch.cuda.synchronize()
    return inp, w, m, d, pc.clone(), pa.clone()


def test_full_batch_sampled_molecules_vs_oracle(full):
    inp, w, m, d, pc, pa = full
    idx = np.random.default_rng(7).choice(4096, size=96, replace=False)
    sub = {k: v[idx] for k, v in inp.items()}
    rc = O.encode(w, "cat", sub["cat_atom"], sub["cat_bond"], sub["cat_connectivity"], pooled_only=True)
    ra = O.encode(w, "an", sub["an_atom"], sub["an_bond"], sub["an_connectivity"], pooled_only=True)
    assert_close(pc.cpu().numpy()[idx], rc, what="cat pooled (sample)")
    assert_close(pa.cpu().numpy()[idx], ra, what="an pooled (sample)")


def test_full_batch_run_to_run_bitwise(full):
    inp, w, m, d, pc, pa = full
    pc2, pa2 = m.encode_pooled(d, fused=True)
    assert torch.equal(pc, pc2) and torch.equal(pa, pa2)


def test_full_batch_shard_concat_bitwise(full):
    """Batch sharding (8e): any contiguous split gives identical rows - samples are independent and the
    kernel's arithmetic per molecule does not depend on its neighbours in the batch."""
    inp, w, m, d, pc, pa = full
    for cuts in ([0, 1000, 4096], [0, 512, 1024, 1536, 2048, 2560, 3072, 3584, 4096]):
        parts_c, parts_a = [], []
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            c, a = m.encode_pooled({k: v[lo:hi] for k, v in d.items()}, fused=True)
            parts_c.append(c); parts_a.append(a)
        assert torch.equal(torch.cat(parts_c), pc) and torch.equal(torch.cat(parts_a), pa)


def test_full_batch_permutation_equivariance(full):
    inp, w, m, d, pc, pa = full
    perm = torch.from_numpy(np.random.default_rng(3).permutation(4096)).to(DEV)
    c, a = m.encode_pooled({k: v[perm].contiguous() for k, v in d.items()}, fused=True)
    assert torch.equal(c, pc[perm]) and torch.equal(a, pa[perm])


def test_full_batch_padding_invariance(full):
    inp, w, m, d, pc, pa = full
    pad = {}
    for k, v in d.items():
        if k == "temperature":
            pad[k] = v
        elif k.endswith("connectivity"):
            pad[k] = torch.cat([v, torch.zeros(v.shape[0], 16, 2, dtype=v.dtype, device=DEV)], 1).contiguous()
        elif k.endswith("bond"):
            pad[k] = torch.cat([v, torch.zeros(v.shape[0], 16, dtype=v.dtype, device=DEV)], 1).contiguous()
        else:
            pad[k] = torch.cat([v, torch.zeros(v.shape[0], 8, dtype=v.dtype, device=DEV)], 1).contiguous()
    c, a = m.encode_pooled(pad, fused=True)
    assert torch.equal(c, pc) and torch.equal(a, pa)


def test_full_batch_fused_equals_layered_hip(full):
    inp, w, m, d, pc, pa = full
    lc, la = m.encode_pooled(d, fused=False)
    assert_close(pc.cpu().numpy(), lc.cpu().numpy(), what="fused vs layered cat")
    assert_close(pa.cpu().numpy(), la.cpu().numpy(), what="fused vs layered an")


@pytest.mark.parametrize("mode", MODES)
def test_prepared_and_per_call_weight_images_agree_bitwise(mode):
    _, inp, w, outs = load_case("config2_perturbed_b6")
    m = make_model(w, 124, 72, mode=mode)
    d = to_dev(inp)
    ions = [(d["cat_atom"], d["cat_bond"], d["cat_connectivity"]), (d["an_atom"], d["an_bond"], d["an_connectivity"])]
    a = ops.encoder_fused(ions, m.atom_emb.embeddings, m.bond_emb.embeddings, m._packed_weights(), m.num_steps, mode=mode)
    b = ops.encoder_fused(ions, m.atom_emb.embeddings, m.bond_emb.embeddings, None, m.num_steps, mode=mode,
                          prepared=m._prepared_weights(mode))
    one = ops.encoder_fused(ions[:1], m.atom_emb.embeddings, m.bond_emb.embeddings, m._packed_weights()[:1], m.num_steps,
                            mode=mode)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(one[0], a[0])
    assert_close(a[0].cpu().numpy(), outs["cat/pooled"], what="cat pooled")


@pytest.mark.parametrize("B", [8192, 70000])
def test_large_batches_shard_consistently(B):
    """Config 3/4 sizes and a batch big enough to need more persistent workgroups than CUs: sampled molecules
    against the oracle, and halves recomputed separately must match bit for bit."""
    inp = synthetic.make_batch(B, seed=12)
    w = weights.init_weights("viscosity", synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, num_steps=2, seed=13, perturb=True)
    m = make_model(w, synthetic.DEFAULT_VA, synthetic.DEFAULT_VB)
    d = to_dev(inp)
    pc, pa = m.encode_pooled(d, fused=True)
    idx = np.random.default_rng(1).choice(B, size=48, replace=False)
    sub = {k: v[idx] for k, v in inp.items()}
    assert_close(pc.cpu().numpy()[idx], O.encode(w, "cat", sub["cat_atom"], sub["cat_bond"], sub["cat_connectivity"],
                                                 pooled_only=True), what="cat pooled (sample)")
    assert_close(pa.cpu().numpy()[idx], O.encode(w, "an", sub["an_atom"], sub["an_bond"], sub["an_connectivity"],
                                                 pooled_only=True), what="an pooled (sample)")
    h = B // 2 + 17
    c0, a0 = m.encode_pooled({k: v[:h].contiguous() for k, v in d.items()}, fused=True)
    c1, a1 = m.encode_pooled({k: v[h:].contiguous() for k, v in d.items()}, fused=True)
    assert torch.equal(torch.cat([c0, c1]), pc) and torch.equal(torch.cat([a0, a1]), pa)


def test_pipelined_plan_run_matches_single_call(full):
    """impnn_encoder_plan on a side stream + impnn_encoder_run == the one-call path, bit for bit, also when
    two planned batches are in flight and run out of order of planning."""
    inp, w, m, d, pc, pa = full
    half = {k: v[:2000].contiguous() for k, v in d.items()}
    h_full = m.plan_batch(d)
    h_half = m.plan_batch(half)
    c2, a2 = m.encode_pooled(half, plan=h_half)
    c1, a1 = m.encode_pooled(d, plan=h_full)
    assert torch.equal(c1, pc) and torch.equal(a1, pa)
    assert torch.equal(c2, pc[:2000]) and torch.equal(a2, pa[:2000])
    for _ in range(5):  # steady-state pipelining loop as in bench.py
        nxt = m.plan_batch(d)
        c, a = m.encode_pooled(d, plan=h_full)
        h_full = nxt
        assert torch.equal(c, pc) and torch.equal(a, pa)


def test_mode_resolution_exact_by_default_split_only_inside_its_range_bound():
    """auto = exact f32 arithmetic ("f32t", else "f32"); "f16x2" runs only on request and only while the
    LayerNorm / in-degree bound keeps every operand inside fp16 range."""
    w = weights.init_weights("viscosity", synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, num_steps=3, seed=1)
    m = make_model(w, synthetic.DEFAULT_VA, synthetic.DEFAULT_VB)
    assert m.resolve_encoder_mode(40, 80) == "f32t"
    assert m.resolve_encoder_mode(40, 300) == "f32t"     # any padded shape since round 3 (per-batch checks in the plan)
    m.encoder_mode = "f16x2"
    assert m.resolve_encoder_mode(40, 80) == "f16x2" and m._split_deg_limit > 80
    big = dict(w)
    big["bond_embedding"] = w["bond_embedding"] * 400.0        # |G| bound explodes -> exact mode
    m.load_weights(big)
    assert m.resolve_encoder_mode(40, 80) == "f32t"
    inp = synthetic.make_batch(16, seed=3)
    ref = O.encode(big, "cat", inp["cat_atom"], inp["cat_bond"], inp["cat_connectivity"], pooled_only=True)
    pc, _ = m.encode_pooled(to_dev(inp), fused=True)
    assert_close(pc.cpu().numpy(), ref, what="exact-mode fallback")
    huge = dict(w)
    huge["cat_bmm_0/bond_transform"] = w["cat_bmm_0/bond_transform"] * 1.0e4   # |W|*256 > fp16 max
    m.load_weights(huge)
    assert m._packed_weights() is not None and m._split_deg_limit == 0.0 and m.resolve_encoder_mode(40, 1) == "f32t"
    # the bf16x9 form on request where the shape fits its LDS budget, the exact-f32 form of the same encoder where not
    m.encoder_mode = "f32x3"
    assert m.resolve_encoder_mode(40, 80) == "f32x3" and m.resolve_encoder_mode(160, 640) == "f32x3"
    assert m.resolve_encoder_mode(300, 3000) == "f32x3"  # (any padded shape: molecules are checked per batch)


def test_plan_and_run_must_agree_on_geometry():
    """impnn_encoder_run refuses a workspace planned for another shape / record kind (host check through the plan
    info), and an encoder kernel that finds a foreign plan header poisons its outputs instead of reading records at
    wrong offsets (device check)."""
    import ctypes as C
    from ionic_mpnn_amd import _lib
    Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
    w = weights.init_weights("viscosity", Va, Vb, num_steps=2, seed=4)
    m = make_model(w, Va, Vb)
    d = to_dev(synthetic.make_batch(300, seed=5))
    ref = m.encode_pooled(d, fused=True)
    m.encoder_workgroups = 64
    plan = m.plan_batch(d)
    assert plan.info.v[8] == 64
    m.encoder_workgroups = 128                      # changing the knob after planning must not matter:
    pc, pa = m.encode_pooled(d, plan=plan)          # the run takes its geometry from the plan
    assert torch.equal(pc, ref[0]) and torch.equal(pa, ref[1])
    plan2 = m.plan_batch(d)
    plan2.mode = "f32"                              # typed records, pull-form kernel: refused on the host
    with pytest.raises(_lib.ImpnnError, match="planned for another"):
        m.encode_pooled(d, plan=plan2)
    plan3 = m.plan_batch(d)
    plan3.shape = (299,) + plan3.shape[1:]          # another batch size
    with pytest.raises(_lib.ImpnnError, match="planned for another"):
        m._pipeline.run(plan3, m.atom_emb.embeddings, m.bond_emb.embeddings, m._prepared_weights("f32t"))
    # device-side check: forge the host info so that the run uses another workgroup count than the plan on the device
    plan4 = m.plan_batch(d)
    torch.cuda.synchronize()
    plan4.info.v[8] = 96
    need = C.c_size_t(0)
    _lib.load().impnn_encoder_workspace_bytes(2, 300, 40, 80, 32, 8, 2, Vb, 2, 96, C.byref(need))
    if plan4.slot["ws"].numel() >= need.value:
        pc, pa = m._pipeline.run(plan4, m.atom_emb.embeddings, m.bond_emb.embeddings, m._prepared_weights("f32t"))
        assert torch.isnan(pc).all() and torch.isnan(pa).all()


@pytest.mark.parametrize("name", ["config2_perturbed_b6", "tiny_melting_point", "tiny_viscosity"])
def test_model_head_kernel_vs_oracle_and_torch_head(name):
    """impnn_model_head (one launch after GlobalSumPool) against the oracle's final output and the torch head."""
    kind, inp, w, outs = load_case(name)
    D, K = w["atom_embedding"].shape[1], w["bond_embedding"].shape[1]
    Va, Vb = w["atom_embedding"].shape[0], w["bond_embedding"].shape[0]
    F, Mx = w["cat_fp/kernel"].shape[1], w["cat_proj/kernel"].shape[1]
    if kind == "viscosity":
        m = MM.build_model(Va, Vb, atom_dim=D, bond_dim=K, fp_size=F, mixing_size=Mx, num_steps=weights.num_steps_of(w), device=DEV)
    else:
        m = MM.build_melting_point_model(Va, Vb, atom_dim=D, fp_size=F, mixing_size=Mx, num_steps=weights.num_steps_of(w), device=DEV)
    m.load_weights(w)
    pc = torch.from_numpy(outs["cat/pooled"].astype(np.float32)).to(DEV)
    pa = torch.from_numpy(outs["an/pooled"].astype(np.float32)).to(DEV)
    T = torch.from_numpy(inp["temperature"]).to(DEV) if kind == "viscosity" else None
    y_kernel = m.head(pc, pa, T)
    y_torch = m.head(pc, pa, T, trace={})
    assert_close(y_kernel.cpu().numpy(), outs["final"], what="head kernel vs oracle")
    assert_close(y_kernel.cpu().numpy(), y_torch.cpu().numpy(), what="head kernel vs torch head")
    assert_close(m(inp).cpu().numpy(), outs["final"], what="full forward")


@pytest.mark.parametrize("kind", ["viscosity", "melting_point"])
def test_model_head_kernel_at_the_widest_head(kind):
    """atom_dim 128 with fp_size = mixing_size = 64: 29 K weight floats, more than the 64 KB default LDS limit holds
    (the head kernel refused what model.head() sent it) - the launch raises its limit to the CU's 160 KB."""
    Va, Vb, B = 20, 6, 37
    w = weights.init_weights(kind, Va, Vb, atom_dim=128, bond_dim=8 if kind == "viscosity" else 128 * 128, fp_size=64,
                             mixing_size=64, num_steps=0, seed=3, perturb=True)
    if kind == "viscosity":
        m = MM.build_model(Va, Vb, atom_dim=128, bond_dim=8, fp_size=64, mixing_size=64, num_steps=0, device=DEV)
    else:
        m = MM.build_melting_point_model(Va, Vb, atom_dim=128, fp_size=64, mixing_size=64, num_steps=0, device=DEV)
    m.load_weights(w)
    rng = np.random.default_rng(0)
    pc = torch.from_numpy(rng.normal(size=(B, 128)).astype(np.float32)).to(DEV)
    pa = torch.from_numpy(rng.normal(size=(B, 128)).astype(np.float32)).to(DEV)
    T = torch.from_numpy(rng.uniform(253, 393, size=(B, 1)).astype(np.float32)).to(DEV) if kind == "viscosity" else None
    y_kernel = m.head(pc, pa, T)                # ops.model_head (one launch)
    y_torch = m.head(pc, pa, T, trace={})       # Dense layers
    assert_close(y_kernel.cpu().numpy(), y_torch.cpu().numpy(), what="widest head: kernel vs torch head")


def test_config1_dataset_plumbing_batch32():
    """BASELINE config 1: records in the *_id_data.pkl schema -> restated loader -> batch 32 -> HIP forward
    vs the oracle."""
    from ionic_mpnn_amd import data
    recs, vocab = synthetic.make_id_records(64, seed=11, max_atoms=14)
    ds = data.IonPairDataset(recs, vocab)
    x = ds.build_inputs(range(32))
    w = weights.init_weights("viscosity", ds.atom_vocab_size, ds.bond_vocab_size, num_steps=4, seed=2)
    m = make_model(w, ds.atom_vocab_size, ds.bond_vocab_size)
    ref = O.viscosity_forward(w, x)
    assert_close(m(x, fused=True).cpu().numpy(), ref, what="log_eta fused")
    assert_close(m(x, fused=False).cpu().numpy(), ref, what="log_eta layered")
    assert_close(m.predict(x, batch_size=32), ref, what="predict")


@pytest.mark.parametrize("mode", MODES)
def test_rows_without_in_edges_never_read_stale_workspace(mode):
    """A chunk whose rows have no in-edge leaves the record's entry area unwritten; the gather of such rows must not
    depend on it (0 * NaN).  The workspace is poisoned with NaN bit patterns first."""
    Va, Vb = 30, 11
    inp = synthetic.make_batch(9, max_atoms=1, max_edges=0, atom_vocab_size=Va, bond_vocab_size=Vb, min_atoms=1, seed=7)
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=32, bond_dim=8, num_steps=2, seed=107, perturb=True)
    m = make_model(w, Va, Vb, K=8, mode=mode)
    from ionic_mpnn_amd import ops
    ws = ops._workspace(torch.device(DEV), 64 << 20)
    ws.view(torch.int32).fill_(0x7FC00000 | 0x3FF)      # quiet NaNs everywhere, offsets fields all ones
    pc, pa = m.encode_pooled(to_dev(inp), fused=True)
    rc = O.encode(w, "cat", inp["cat_atom"], inp["cat_bond"], inp["cat_connectivity"], pooled_only=True)
    assert torch.isfinite(pc).all() and torch.isfinite(pa).all()
    assert_close(pc.cpu().numpy(), rc, what="cat pooled")


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("Va,S", [(600, 3), (455, 1), (454, 2), (600, 0)])
def test_large_atom_vocabularies_take_the_global_table_path(Va, S, mode):
    """Up to 454 atom types the embedding table is copied to LDS and step 0 gathers from it; above that the chunk
    prologue fills h0 from HBM and step 0 gathers from the h buffer (with the mid-step barrier).  Both must match."""
    Vb = 9
    inp = synthetic.make_batch(70, max_atoms=30, max_edges=50, atom_vocab_size=Va, bond_vocab_size=Vb, min_atoms=4, seed=Va + S)
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=32, bond_dim=8, num_steps=S, seed=Va, perturb=True)
    m = make_model(w, Va, Vb, K=8, mode=mode)
    pc, pa = m.encode_pooled(to_dev(inp), fused=True)
    rc = O.encode(w, "cat", inp["cat_atom"], inp["cat_bond"], inp["cat_connectivity"], pooled_only=True)
    ra = O.encode(w, "an", inp["an_atom"], inp["an_bond"], inp["an_connectivity"], pooled_only=True)
    assert_close(pc.cpu().numpy(), rc, what="cat pooled")
    assert_close(pa.cpu().numpy(), ra, what="an pooled")


def _random_dense_case(rng):
    N = int(rng.integers(1, 97))
    E = int(rng.integers(0, 4 * N + 1))
    E = min(E, 300)
    K = int(rng.integers(1, 9))
    S = int(rng.integers(0, 5))
    B = int(rng.integers(1, 400))
    Va, Vb = int(rng.integers(2, 500)), int(rng.integers(1, 200))
    ids = rng.integers(0, Va, size=(2, B, N)).astype(np.int32)
    ids[rng.random(size=ids.shape) < 0.2] = 0                      # padding holes anywhere
    conn = rng.integers(0, N, size=(2, B, E, 2)).astype(np.int32)   # arbitrary multigraph incl. self loops
    bond = rng.integers(0, Vb, size=(2, B, E)).astype(np.int32)
    inp = {"cat_atom": ids[0], "cat_bond": bond[0], "cat_connectivity": conn[0],
           "an_atom": ids[1], "an_bond": bond[1], "an_connectivity": conn[1]}
    return N, E, K, S, B, Va, Vb, inp


@pytest.mark.parametrize("seed", range(24))
def test_fused_encoder_fuzz_against_the_oracle(seed):
    """Random shapes and arbitrary multigraphs (not just trees): every placement / chunking / entry-packing path of
    the plan and both table paths of the encoder, in the mode the model would pick."""
    rng = np.random.default_rng(1000 + seed)
    N, E, K, S, B, Va, Vb, inp = _random_dense_case(rng)
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=32, bond_dim=K, num_steps=S, seed=seed, perturb=True)
    m = make_model(w, Va, Vb, K=K, mode="auto")
    if not m.fused_supported(N, E):
        pytest.skip("shape outside the fused encoder")
    pc, pa = m.encode_pooled(to_dev(inp), fused=True)
    rc = O.encode(w, "cat", inp["cat_atom"], inp["cat_bond"], inp["cat_connectivity"], pooled_only=True)
    ra = O.encode(w, "an", inp["an_atom"], inp["an_bond"], inp["an_connectivity"], pooled_only=True)
    assert_close(pc.cpu().numpy(), rc, what=f"cat pooled (N={N} E={E} K={K} S={S} B={B} Va={Va} Vb={Vb})")
    assert_close(pa.cpu().numpy(), ra, what="an pooled")


def test_batches_in_flight_on_two_streams_do_not_share_a_workspace():
    """bench.py sends consecutive batches to two HIP streams: every stream has its own encoder workspace
    (ops._workspace), so two different batches in flight give exactly the results of running them one after the other."""
    Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
    m = MM.build_model(Va, Vb, device=DEV)
    m.load_weights(weights.init_weights("viscosity", Va, Vb, seed=3, perturb=True))
    batches = [{k: torch.from_numpy(v).to(DEV) for k, v in synthetic.make_batch(1536, seed=s).items()} for s in (1, 2)]
    ref = [tuple(t.clone() for t in m.encode_pooled(b, fused=True)) for b in batches]
    torch.cuda.synchronize()
    lanes = [torch.cuda.Stream(device=DEV) for _ in range(2)]
    for rep in range(20):
        outs = []
        for i in (0, 1):
            with torch.cuda.stream(lanes[i]):
                outs.append(m.encode_pooled(batches[i], fused=True))
        torch.cuda.synchronize()
        for i in (0, 1):
            assert torch.equal(outs[i][0], ref[i][0]) and torch.equal(outs[i][1], ref[i][1])
    assert len({k for k in ops._workspaces if k[1] in (lanes[0].cuda_stream, lanes[1].cuda_stream)}) == 2
    # fewer persistent workgroups per launch (what bench.py uses with three streams): same results bit for bit
    for wgs in (128, 48, 1000):
        m.encoder_workgroups = wgs
        for i in (0, 1):
            pc, pa = m.encode_pooled(batches[i], fused=True)
            assert torch.equal(pc, ref[i][0]) and torch.equal(pa, ref[i][1])
    m.encoder_workgroups = 0


@pytest.mark.parametrize("fused", [True, False])
def test_predict_in_chunks_equals_predict_at_once(fused):
    """model.predict(x, batch_size) sends consecutive chunks to two streams and synchronises once."""
    Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
    m = MM.build_model(Va, Vb, device=DEV)
    m.load_weights(weights.init_weights("viscosity", Va, Vb, seed=5, perturb=True))
    inp = synthetic.make_batch(203, seed=11)
    whole = m.predict(inp, fused=fused)
    for bs in (32, 100):
        parts = m.predict(inp, batch_size=bs, fused=fused)
        assert parts.shape == whole.shape
        np.testing.assert_allclose(parts, whole, rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------ BASELINE.json configs 3 and 5 at their real shapes
@pytest.fixture(scope="module")
def config3():
    """Melting-point model at its real shape (train_melting_point.py:137-198): D=32, K=D*D=1024, S=4, batch 8192
    (BASELINE.json configs[2]; N=40 / E=80 / vocabulary stand-ins as SURVEY.md 8(d) - the real pkl is absent)."""
    Va, Vb, B = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, 8192
    inp = synthetic.make_batch(B, seed=21, with_temperature=False)
    w = weights.init_weights("melting_point", Va, Vb, atom_dim=32, bond_dim=1024, num_steps=4, seed=22, perturb=True)
    m = MM.build_melting_point_model(Va, Vb, atom_dim=32, num_steps=4, device=DEV)
    m.load_weights(w)
    d = to_dev(inp)
    return inp, w, m, d


def test_config3_melting_point_real_shape_fused_typed(config3):
    inp, w, m, d = config3
    assert m.bond_dim == 1024 and m.resolve_encoder_mode(40, 80) == "f32t"   # K = D^2 runs in the fused encoder
    pc, pa = m.encode_pooled(d, fused=True)
    y = m(d, fused=True)
    torch.cuda.synchronize()
    idx = np.random.default_rng(5).choice(8192, size=40, replace=False)
    sub = {k: v[idx] for k, v in inp.items()}
    rc = O.encode(w, "cat", sub["cat_atom"], sub["cat_bond"], sub["cat_connectivity"], pooled_only=True)
    ra = O.encode(w, "an", sub["an_atom"], sub["an_bond"], sub["an_connectivity"], pooled_only=True)
    assert_close(pc.cpu().numpy()[idx], rc, what="config 3 cat pooled (sample)")
    assert_close(pa.cpu().numpy()[idx], ra, what="config 3 an pooled (sample)")
    assert_close(y.cpu().numpy()[idx], O.melting_point_forward(w, sub), what="config 3 melting-point head (sample)")
    # batch shards recomputed separately are the same rows bit for bit
    h = 8192 // 2 + 33
    c0, a0 = m.encode_pooled({k: v[:h].contiguous() for k, v in d.items()}, fused=True)
    c1, a1 = m.encode_pooled({k: v[h:].contiguous() for k, v in d.items()}, fused=True)
    assert torch.equal(torch.cat([c0, c1]), pc) and torch.equal(torch.cat([a0, a1]), pa)
    # and the layer-at-a-time HIP path (per-bond-type matrices through impnn_bond_type_matrices) agrees
    lc, la = m.encode_pooled({k: v[:512].contiguous() for k, v in d.items()}, fused=False)
    assert_close(lc.cpu().numpy(), pc[:512].cpu().numpy(), what="config 3 layered vs fused cat")
    assert_close(la.cpu().numpy(), pa[:512].cpu().numpy(), what="config 3 layered vs fused an")


def test_config5_shape_forward_vs_oracle():
    """BASELINE.json configs[4]'s forward shape: atom_dim 128, 6 message-passing steps, batch 4096 (the validation /
    predict path of the full training loop), sampled molecules against the oracle."""
    Va, Vb, B = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, 4096
    inp = synthetic.make_batch(B, seed=31)
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=128, bond_dim=8, num_steps=6, seed=32, perturb=True)
    m = MM.build_model(Va, Vb, atom_dim=128, bond_dim=8, num_steps=6, device=DEV)
    m.load_weights(w)
    d = to_dev(inp)
    pc, pa = m.encode_pooled(d)
    y = m(d)
    torch.cuda.synchronize()
    idx = np.random.default_rng(6).choice(B, size=24, replace=False)
    sub = {k: v[idx] for k, v in inp.items()}
    rc = O.encode(w, "cat", sub["cat_atom"], sub["cat_bond"], sub["cat_connectivity"], pooled_only=True)
    ra = O.encode(w, "an", sub["an_atom"], sub["an_bond"], sub["an_connectivity"], pooled_only=True)
    assert_close(pc.cpu().numpy()[idx], rc, what="config 5 cat pooled (sample)")
    assert_close(pa.cpu().numpy()[idx], ra, what="config 5 an pooled (sample)")
    assert_close(y.cpu().numpy()[idx], O.viscosity_forward(w, sub), what="config 5 log_eta (sample)")
    h = B // 2 - 5
    c0, a0 = m.encode_pooled({k: v[:h].contiguous() for k, v in d.items()})
    c1, a1 = m.encode_pooled({k: v[h:].contiguous() for k, v in d.items()})
    assert torch.equal(torch.cat([c0, c1]), pc) and torch.equal(torch.cat([a0, a1]), pa)


@pytest.mark.parametrize("K,Vb", [(1024, 72), (64, 200), (17, 256), (1, 3)])
def test_typed_encoder_any_bond_dim(K, Vb):
    """The typed mode's only use of bond_dim is the prepared per-bond-type matrices: any K, up to 256 bond types."""
    Va = 50
    inp = synthetic.make_batch(130, max_atoms=30, max_edges=60, atom_vocab_size=Va, bond_vocab_size=Vb, min_atoms=4,
                               seed=K, with_temperature=False)
    w = weights.init_weights("melting_point", Va, Vb, atom_dim=32, bond_dim=K, num_steps=2, seed=K + 1, perturb=True)
    m = MM.MPNNModel("melting_point", Va, Vb, 32, K, 32, 20, 2, 1e-4, device=DEV)
    m.load_weights(w)
    m.encoder_mode = "f32t"
    pc, pa = m.encode_pooled(to_dev(inp), fused=True)
    rc = O.encode(w, "cat", inp["cat_atom"], inp["cat_bond"], inp["cat_connectivity"], pooled_only=True)
    ra = O.encode(w, "an", inp["an_atom"], inp["an_bond"], inp["an_connectivity"], pooled_only=True)
    assert_close(pc.cpu().numpy(), rc, what=f"cat pooled K={K} Vb={Vb}")
    assert_close(pa.cpu().numpy(), ra, what="an pooled")


def test_f32x3_is_as_accurate_as_the_f32_mfma():
    """Mode "f32x3" carries every f32 operand of the GatedUpdate GEMMs as an exact sum of three bf16 terms and keeps all
    nine cross products, so its error against the fp64 oracle must be of the size of the exact-f32 mode's own
    (f32 accumulation in another order) - measured elementwise, not only against the tensor's scale."""
    Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
    inp = synthetic.make_batch(384, seed=41)
    w = weights.init_weights("viscosity", Va, Vb, num_steps=4, seed=42, perturb=True)
    ref = np.concatenate([O.encode(w, p, inp[f"{p}_atom"], inp[f"{p}_bond"], inp[f"{p}_connectivity"], pooled_only=True)
                          for p in ("cat", "an")])
    err = {}
    for mode in ("f32t", "f32x3"):
        m = make_model(w, Va, Vb, mode=mode)
        pc, pa = m.encode_pooled(to_dev(inp), fused=True)
        got = np.concatenate([pc.cpu().numpy(), pa.cpu().numpy()]).astype(np.float64)
        d = np.abs(got - ref)
        err[mode] = (float(d.max() / np.abs(ref).max()), float(np.sqrt(np.mean(d * d)) / np.sqrt(np.mean(ref * ref))),
                     float(np.max(d / np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max()))))
    for i in range(3):
        assert err["f32x3"][i] <= 2.0 * err["f32t"][i] + 1e-7, err
    assert err["f32x3"][0] <= 1e-5 and err["f32t"][0] <= 1e-5


def _mode_errors(w, inp, Va, Vb, K=8, sample=None, kind="viscosity"):
    """max / rms / elementwise error of modes f32t and f32x3 against the fp64 oracle (relative to the tensor's scale)."""
    sub = inp if sample is None else {k: v[sample] for k, v in inp.items()}
    ref = np.concatenate([O.encode(w, p, sub[f"{p}_atom"], sub[f"{p}_bond"], sub[f"{p}_connectivity"], pooled_only=True)
                          for p in ("cat", "an")])
    d_in, err, out = to_dev(inp), {}, {}
    for mode in ("f32t", "f32x3"):
        if kind == "viscosity":
            m = make_model(w, Va, Vb, K=K, mode=mode)
        else:
            m = MM.build_melting_point_model(Va, Vb, atom_dim=32, num_steps=weights.num_steps_of(w), device=DEV)
            m.load_weights(w)
            m.encoder_mode = mode
        assert m.bond_dim == K and m.resolve_encoder_mode(inp["cat_atom"].shape[1], inp["cat_bond"].shape[1]) == mode
        pc, pa = m.encode_pooled(d_in, fused=True)
        pc, pa = pc.cpu().numpy(), pa.cpu().numpy()
        if sample is not None:
            pc, pa = pc[sample], pa[sample]
        got = np.concatenate([pc, pa]).astype(np.float64)
        out[mode] = got
        scale = np.abs(ref).max()
        with np.errstate(invalid="ignore"):
            d = np.abs(got - ref)
        err[mode] = (float(d.max() / scale), float(np.sqrt(np.mean(d * d)) / np.sqrt(np.mean(ref * ref))),
                     float(np.max(d / np.maximum(np.abs(ref), 1e-3 * scale))))
    return err, out, ref


@pytest.mark.parametrize("config", ["config2", "config3"])
def test_f32x3_error_within_twice_f32t_on_the_baseline_configs(config):
    """VERDICT r2, condition (b) for mode f32x3 ("f32 (bf16x9 emulation)"): on BASELINE configs 2 and 3 (full batch; the
    oracle on a sample) its error against the fp64 oracle is at most twice the exact-f32 mode's.  (Config 5 is atom_dim
    128: the wide encoder, which has no such mode.)"""
    Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
    if config == "config2":
        B, K, S, kind = 4096, 8, 3, "viscosity"
    else:
        B, K, S, kind = 8192, 1024, 4, "melting_point"
    inp = synthetic.make_batch(B, seed=61, with_temperature=False)
    w = weights.init_weights(kind, Va, Vb, atom_dim=32, bond_dim=K, num_steps=S, seed=62, perturb=True)
    idx = np.random.default_rng(8).choice(B, size=96, replace=False)
    err, _, _ = _mode_errors(w, inp, Va, Vb, K=K, sample=idx, kind=kind)
    for i in range(3):
        assert err["f32x3"][i] <= 2.0 * err["f32t"][i] + 1e-7, err
    assert err["f32x3"][0] <= 1e-5 and err["f32t"][0] <= 1e-5, err


@pytest.mark.parametrize("what", ["embeddings", "bond_transform", "gate_kernels", "everything"])
def test_f32x3_magnitude_sweep(what):
    """The same bound over forty orders of magnitude: atom / bond embeddings, the message weights and the GatedUpdate
    kernels scaled by 1e-20 ... 1e+20 (bf16 keeps fp32's exponent, so the three-term split is exact wherever the f32
    value is normal; the third term reaches bf16's subnormals only for |x| < ~1e-33)."""
    Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
    inp = synthetic.make_batch(160, seed=71, with_temperature=False)
    w0 = weights.init_weights("viscosity", Va, Vb, num_steps=3, seed=72, perturb=True)
    for s in (1e-20, 1e-12, 1e-6, 1e-3, 1.0, 1e3, 1e6, 1e12, 1e20):
        w = dict(w0)
        for k in w0:
            hit = (what in ("embeddings", "everything") and k in ("atom_embedding", "bond_embedding")) or \
                  (what in ("bond_transform", "everything") and k.endswith("bond_transform")) or \
                  (what in ("gate_kernels", "everything") and "/dense_" in k and k.endswith("kernel") and "_gu_" in k)
            if hit:
                sk = s if what != "everything" else (s if k.endswith("embedding") else (1.0 / s if "dense" in k else np.sqrt(s)))
                w[k] = (w0[k].astype(np.float64) * sk).astype(np.float32)
        err, out, ref = _mode_errors(w, inp, Va, Vb)
        # (at the far ends f32 itself leaves its range - LayerNorm's sum of 32 squares of 1e19 overflows - for either
        #  mode alike: what is required is the same finiteness pattern and the 2x bound, and 1e-5 wherever f32t has it)
        assert np.isfinite(ref).all(), (what, s)
        assert np.array_equal(np.isfinite(out["f32t"]), np.isfinite(out["f32x3"])), (what, s)
        if np.isfinite(out["f32t"]).all():
            if err["f32t"][0] <= 1e-5:   # wherever exact f32 meets the path's tolerance: within twice its error, and 1e-5
                for i in range(3):
                    assert err["f32x3"][i] <= 2.0 * err["f32t"][i] + 1e-7, (what, s, err)
                assert err["f32x3"][0] <= 1e-5, (what, s, err)
            else:
                # ill-conditioned corners (message weights x 1e6: saturated gates amplify ANY f32 rounding ~100x; exact
                # f32 itself is at 1e-4 there): the two modes differ by their summation order only, i.e. by another
                # draw of the same rounding noise - the same order of magnitude is what can be asked
                for i in range(3):
                    assert err["f32x3"][i] <= 4.0 * err["f32t"][i] + 1e-7, (what, s, err)


def test_f32x3_propagates_nan_and_inf_like_f32t():
    """A NaN in an embedding row or a weight makes the SAME molecules' outputs NaN in both modes, every other
    molecule's output is untouched.  An infinity: exact f32 has w * inf = +-inf, which a saturating gate may turn back
    into a finite number; the three-term split turns an infinite operand into (inf, nan, nan) - mode f32x3 is the more
    conservative one: its non-finite rows are a superset of f32t's, never a finite value where f32t reports none, and
    the rows it does report finite agree with f32t's."""
    Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
    inp = synthetic.make_batch(200, seed=81, with_temperature=False)
    w0 = weights.init_weights("viscosity", Va, Vb, num_steps=3, seed=82, perturb=True)
    clean = {}
    for mode in ("f32t", "f32x3"):
        pc, pa = make_model(w0, Va, Vb, mode=mode).encode_pooled(to_dev(inp), fused=True)
        clean[mode] = torch.cat([pc, pa]).cpu().numpy()
    for bad in (np.nan, np.inf, -np.inf):
        for key, where in (("atom_embedding", (17, 5)), ("bond_embedding", (9, 2)), ("cat_gu_1/dense_r/kernel", (40, 3)),
                           ("an_bmm_2/bond_transform", (3, 7, 11))):
            w = {k: v.copy() for k, v in w0.items()}
            w[key][where] = bad
            res = {}
            for mode in ("f32t", "f32x3"):
                pc, pa = make_model(w, Va, Vb, mode=mode).encode_pooled(to_dev(inp), fused=True)
                res[mode] = torch.cat([pc, pa]).cpu().numpy()
            fin_t, fin_x = np.isfinite(res["f32t"]).all(axis=1), np.isfinite(res["f32x3"]).all(axis=1)
            assert not fin_t.all(), (bad, key)   # the poison reached some molecule
            assert not (fin_x & ~fin_t).any(), (bad, key)   # never finite where exact f32 is not
            if np.isnan(bad):
                assert np.array_equal(fin_t, fin_x), (bad, key, int(fin_t.sum()), int(fin_x.sum()))
                assert np.array_equal(np.isnan(res["f32t"]), np.isnan(res["f32x3"])), (bad, key)
            both = fin_t & fin_x
            if both.any():
                assert_close(res["f32x3"][both], res["f32t"][both].astype(np.float64), what=f"finite rows ({bad}, {key})")
            if key.endswith("embedding"):        # only molecules that hold the poisoned id: the others are bit-identical
                assert fin_t.any()
                for mode, fin in (("f32t", fin_t), ("f32x3", fin_x)):
                    touched = np.any(res[mode] != clean[mode], axis=1) | ~fin
                    assert np.array_equal(res[mode][~touched], clean[mode][~touched]) and (~touched).any(), (mode, bad, key)


# ---------------------------------------------------------------------------------------------------------------
# round 3: the padded shapes of the reference's real (explicit-hydrogen) data sets.  featurize.py:45 adds hydrogens and
# the trainers pad to E = 4 * max_bonds edge slots (train_viscosity.py:95,288-289): N = 160, E = 640 here.  The typed
# encoder bounds a chunk by what a molecule HOLDS (plan kernels, per batch), not by the padded shape.
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mode", ["auto", "f32x3"])
@pytest.mark.parametrize("kind,K,S", [("viscosity", 8, 3), ("melting_point", 1024, 4)])
def test_explicit_h_shape_runs_in_the_typed_encoder(kind, K, S, mode):
    Va, Vb, B, N, E = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, 4096, 160, 640
    inp = synthetic.make_explicit_h_batch(B, max_atoms=N, max_edges=E, seed=41, with_temperature=(kind == "viscosity"))
    valid = (inp["cat_connectivity"][:, :, 0] > 0) & (inp["cat_connectivity"][:, :, 1] > 0)
    assert valid.sum(axis=1).max() > 512 and (inp["cat_atom"] > 0).sum(axis=1).max() > 128  # beyond round 2's limits
    w = weights.init_weights(kind, Va, Vb, atom_dim=32, bond_dim=K, num_steps=S, seed=42, perturb=True)
    if kind == "viscosity":
        m = MM.build_model(Va, Vb, atom_dim=32, bond_dim=K, num_steps=S, device=DEV)
    else:
        m = MM.build_melting_point_model(Va, Vb, atom_dim=32, num_steps=S, device=DEV)   # bond_dim = 32 ** 2
    m.load_weights(w)
    m.encoder_mode = mode  # (f32x3 on 640-edge chunks: the kernel instantiation with unpadded h rows)
    assert m.bond_dim == K and m.resolve_encoder_mode(N, E) == ("f32t" if mode == "auto" else mode)
    d = to_dev(inp)
    pc, pa = m.encode_pooled(d, fused=True)
    torch.cuda.synchronize()
    assert getattr(m, "overflow_fallbacks", 0) == 0          # every molecule fits a chunk: the fused kernel ran
    assert torch.isfinite(pc).all() and torch.isfinite(pa).all()
    idx = np.random.default_rng(7).choice(B, size=20, replace=False)
    idx[0] = int(valid.sum(axis=1).argmax())                   # the largest cation is in the sample
    sub = {k: v[idx] for k, v in inp.items()}
    rc = O.encode(w, "cat", sub["cat_atom"], sub["cat_bond"], sub["cat_connectivity"], pooled_only=True)
    ra = O.encode(w, "an", sub["an_atom"], sub["an_bond"], sub["an_connectivity"], pooled_only=True)
    assert_close(pc.cpu().numpy()[idx], rc, what="explicit-H cat pooled (sample)")
    assert_close(pa.cpu().numpy()[idx], ra, what="explicit-H an pooled (sample)")
    # batch shards recomputed separately are the same rows bit for bit
    h = B // 2 + 17
    c0, a0 = m.encode_pooled({k: v[:h].contiguous() for k, v in d.items()}, fused=True)
    c1, a1 = m.encode_pooled({k: v[h:].contiguous() for k, v in d.items()}, fused=True)
    assert torch.equal(torch.cat([c0, c1]), pc) and torch.equal(torch.cat([a0, a1]), pa)
    # and the layer-at-a-time HIP path agrees
    lc, la = m.encode_pooled({k: v[:64].contiguous() for k, v in d.items()}, fused=False)
    assert_close(lc.cpu().numpy(), pc[:64].cpu().numpy(), what="explicit-H layered vs fused cat")
    assert_close(la.cpu().numpy(), pa[:64].cpu().numpy(), what="explicit-H layered vs fused an")


def test_molecule_beyond_a_chunk_is_flagged_and_falls_back():
    """A molecule that truly exceeds a chunk (here: 700 valid edges in E = 800 slots; 300 kept rows) makes the plan raise
    its overflow word: the fused entry returns NaN + EncoderOverflow, the model takes the layer-at-a-time path for
    that batch and the answer is the oracle's."""
    Va, Vb, B = 30, 11, 12
    for N, E, fat in ((200, 800, "edges"), (300, 400, "rows"), (40, 600, "indegree")):
        inp = synthetic.make_batch(B, max_atoms=min(N, 60), max_edges=80, atom_vocab_size=Va, bond_vocab_size=Vb, seed=5)
        pad = lambda a, shape: np.pad(a, [(0, t - s_) for s_, t in zip(a.shape, shape)])
        inp = {k: (pad(v, (B, N)) if k.endswith("atom") else pad(v, (B, E)) if k.endswith("bond") else
                   pad(v, (B, E, 2)) if k.endswith("connectivity") else v) for k, v in inp.items()}
        rng = np.random.default_rng(9)
        if fat == "edges":      # molecule 3 of the cation: 700 valid edges over 180 atoms
            n = 180
            inp["cat_atom"][3, :n] = rng.integers(1, Va, size=n)
            src = rng.integers(1, n, size=700)
            tgt = 1 + (src + rng.integers(0, n - 2, size=700)) % (n - 1)
            inp["cat_connectivity"][3, :700, 0], inp["cat_connectivity"][3, :700, 1] = src, tgt
            inp["cat_bond"][3, :700] = rng.integers(1, Vb, size=700)
        elif fat == "rows":     # molecule 5 of the anion: 300 atoms in a chain
            inp["an_atom"][5, :] = rng.integers(1, Va, size=N)
            k = np.arange(1, 200)
            inp["an_connectivity"][5, :199, 0], inp["an_connectivity"][5, :199, 1] = k, k + 1
            inp["an_bond"][5, :199] = rng.integers(1, Vb, size=199)
        else:                   # 300 edges into one atom: in-degree beyond the 8-bit field
            inp["cat_atom"][2, :40] = rng.integers(1, Va, size=40)
            inp["cat_connectivity"][2, :300, 0] = rng.integers(1, 40, size=300)
            inp["cat_connectivity"][2, :300, 1] = 7
            inp["cat_bond"][2, :300] = rng.integers(1, Vb, size=300)
        w = weights.init_weights("viscosity", Va, Vb, atom_dim=32, bond_dim=8, num_steps=2, seed=77, perturb=True)
        m = make_model(w, Va, Vb, mode="f32t")
        d = to_dev(inp)
        with pytest.raises(ops.EncoderOverflow):
            ops.encoder_fused([(d["cat_atom"], d["cat_bond"], d["cat_connectivity"]),
                               (d["an_atom"], d["an_bond"], d["an_connectivity"])], m.atom_emb.embeddings,
                              m.bond_emb.embeddings, None, 2, mode="f32t", prepared=m._prepared_weights("f32t"))
        pc, pa = m.encode_pooled(d)          # fused by default -> overflow -> layered
        assert m.overflow_fallbacks == 1, fat
        rc = O.encode(w, "cat", inp["cat_atom"], inp["cat_bond"], inp["cat_connectivity"], pooled_only=True)
        ra = O.encode(w, "an", inp["an_atom"], inp["an_bond"], inp["an_connectivity"], pooled_only=True)
        assert_close(pc.cpu().numpy(), rc, what=f"fallback cat pooled ({fat})")
        assert_close(pa.cpu().numpy(), ra, what=f"fallback an pooled ({fat})")
        # the same shapes WITHOUT the fat molecule run fused
        for k in ("cat", "an"):
            for r in (2, 3, 5):
                inp[f"{k}_atom"][r] = 0
                inp[f"{k}_bond"][r] = 0
                inp[f"{k}_connectivity"][r] = 0
        m.overflow_fallbacks = 0
        pc, _ = m.encode_pooled(to_dev(inp))
        assert m.overflow_fallbacks == 0 and torch.isfinite(pc).all()
        assert_close(pc.cpu().numpy(), O.encode(w, "cat", inp["cat_atom"], inp["cat_bond"], inp["cat_connectivity"],
                                                pooled_only=True), what=f"fused after removing the fat molecule ({fat})")
