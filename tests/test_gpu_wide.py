"""Wide atom states (atom_dim 64 / 128) through the encoder entries: csrc/encoder_wide.hip against the oracle
(oracle/mpnn_oracle.py, the restatement of models/layers.py:57-164 and train_viscosity.py:166-190) and against the
layer-at-a-time HIP kernels.  Tolerance 1e-5 relative (BASELINE.json north_star), conftest.assert_close."""
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


def make_model(w, Va, Vb, D, K=8, mode="auto"):
    m = MM.build_model(Va, Vb, atom_dim=D, bond_dim=K, num_steps=weights.num_steps_of(w), device=DEV)
    m.load_weights(w)
    m.encoder_mode = mode
    return m


WIDE_MODES = ["f32t", "f32x3"]  # exact f32 MFMA; the GEMMs of the message (atom_dim 128) and GatedUpdate layers as bf16x9 emulation


def oracle_pooled(w, inp):
    return (O.encode(w, "cat", inp["cat_atom"], inp["cat_bond"], inp["cat_connectivity"], pooled_only=True),
            O.encode(w, "an", inp["an_atom"], inp["an_bond"], inp["an_connectivity"], pooled_only=True))


@pytest.mark.parametrize("D,N,E,K,S,B,seed", [(128, 40, 80, 8, 3, 37, 1), (128, 40, 80, 8, 6, 130, 2),
                                              (64, 40, 80, 8, 3, 65, 3), (64, 12, 20, 4, 2, 100, 4),
                                              (128, 7, 30, 1, 1, 50, 5), (128, 40, 80, 5, 0, 30, 6),
                                              (64, 1, 0, 8, 2, 9, 7), (128, 100, 240, 8, 1, 16, 8),
                                              (64, 128, 512, 3, 1, 5, 9), (128, 256, 40, 2, 2, 3, 10)])
@pytest.mark.parametrize("mode", WIDE_MODES)
def test_wide_encoder_random_shapes(D, N, E, K, S, B, seed, mode):
    Va, Vb = 30, 11
    inp = synthetic.make_batch(B, max_atoms=N, max_edges=E, atom_vocab_size=Va, bond_vocab_size=Vb,
                               min_atoms=min(3, N), seed=seed)
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=D, bond_dim=K, num_steps=S, seed=seed + 100, perturb=True)
    m = make_model(w, Va, Vb, D, K)
    assert m.resolve_encoder_mode(N, E) == "f32t"       # auto = exact f32
    m.encoder_mode = mode
    assert m.resolve_encoder_mode(N, E) == mode
    pc, pa = m.encode_pooled(to_dev(inp), fused=True)
    rc, ra = oracle_pooled(w, inp)
    assert_close(pc.cpu().numpy(), rc, what="cat pooled")
    assert_close(pa.cpu().numpy(), ra, what="an pooled")


@pytest.mark.parametrize("mode", WIDE_MODES)
@pytest.mark.parametrize("D", [64, 128])
def test_wide_encoder_adversarial_graphs(D, mode):
    """Edges that name padding atoms, self loops, 4x duplicated bonds (trainer expansion), id-0 holes, all-padding
    molecules, one atom with 64 in-edges, bond ids outside the vocabulary - the general contract."""
    rng = np.random.default_rng(42)
    B, N, E, Va, Vb, K, S = 48, 24, 64, 20, 7, 8, 3
    ids = rng.integers(0, Va, size=(B, N)).astype(np.int32)
    ids[0] = 0
    conn = rng.integers(0, N, size=(B, E, 2)).astype(np.int32)
    conn[1] = 5
    conn[2, :, 1] = 3
    conn[:, 48:] = 0
    bond = rng.integers(0, Vb, size=(B, E)).astype(np.int32)
    e4, b4 = O.preprocess_edges_and_bonds([[(0, 1), (1, 0), (1, 2), (2, 1), (2, 3), (3, 2)]] * 4,
                                          [[1, 1, 2, 2, 3, 3]] * 4, E // 2)
    conn[3:7], bond[3:7] = e4, b4
    inp = {"cat_atom": ids, "cat_bond": bond, "cat_connectivity": conn,
           "an_atom": ids[::-1].copy(), "an_bond": bond[::-1].copy(), "an_connectivity": conn[::-1].copy()}
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=D, bond_dim=K, num_steps=S, seed=9, perturb=True)
    m = make_model(w, Va, Vb, D, K, mode=mode)
    pc, pa = m.encode_pooled(to_dev(inp), fused=True)
    rc, ra = oracle_pooled(w, inp)
    assert_close(pc.cpu().numpy(), rc, what="cat pooled")
    assert_close(pa.cpu().numpy(), ra, what="an pooled")
    assert float(pc[0].abs().max()) == 0.0


@pytest.fixture(scope="module")
def wide_full():
    """BASELINE.json configs[4]'s forward shape at a size the module can afford several times: D=128, S=6."""
    Va, Vb, B = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, 1024
    inp = synthetic.make_batch(B, seed=41)
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=128, bond_dim=8, num_steps=6, seed=42, perturb=True)
    m = make_model(w, Va, Vb, 128)
    d = to_dev(inp)
    pc, pa = m.encode_pooled(d, fused=True)
    torch.cuda.synchronize()
    return inp, w, m, d, pc.clone(), pa.clone()


def test_wide_sampled_molecules_vs_oracle(wide_full):
    inp, w, m, d, pc, pa = wide_full
    idx = np.random.default_rng(3).choice(1024, size=20, replace=False)
    sub = {k: v[idx] for k, v in inp.items()}
    rc, ra = oracle_pooled(w, sub)
    assert_close(pc.cpu().numpy()[idx], rc, what="cat pooled (sample)")
    assert_close(pa.cpu().numpy()[idx], ra, what="an pooled (sample)")
    y = m(d).cpu().numpy()[idx]
    assert_close(y, O.viscosity_forward(w, sub), what="log_eta (sample)")


def test_wide_run_to_run_and_shard_concat_bitwise(wide_full):
    inp, w, m, d, pc, pa = wide_full
    c2, a2 = m.encode_pooled(d, fused=True)
    assert torch.equal(c2, pc) and torch.equal(a2, pa)
    h = 1024 // 2 - 7
    c0, a0 = m.encode_pooled({k: v[:h].contiguous() for k, v in d.items()}, fused=True)
    c1, a1 = m.encode_pooled({k: v[h:].contiguous() for k, v in d.items()}, fused=True)
    assert torch.equal(torch.cat([c0, c1]), pc) and torch.equal(torch.cat([a0, a1]), pa)


def test_wide_permutation_and_padding_invariance(wide_full):
    inp, w, m, d, pc, pa = wide_full
    perm = torch.from_numpy(np.random.default_rng(5).permutation(1024)).to(DEV)
    cp, ap = m.encode_pooled({k: v[perm].contiguous() for k, v in d.items()}, fused=True)
    assert torch.equal(cp, pc[perm]) and torch.equal(ap, pa[perm])
    # more padding (N 40 -> 56, E 80 -> 100) changes nothing, bit for bit
    wide = {}
    for k, v in d.items():
        if k.endswith("_atom"):
            wide[k] = torch.nn.functional.pad(v, (0, 16))
        elif k.endswith("_bond"):
            wide[k] = torch.nn.functional.pad(v, (0, 20))
        elif k.endswith("_connectivity"):
            wide[k] = torch.nn.functional.pad(v, (0, 0, 0, 20))
        else:
            wide[k] = v
    cw, aw = m.encode_pooled(wide, fused=True)
    assert torch.equal(cw, pc) and torch.equal(aw, pa)


def test_wide_equals_layered_hip(wide_full):
    inp, w, m, d, pc, pa = wide_full
    sub = {k: v[:256].contiguous() for k, v in d.items()}
    lc, la = m.encode_pooled(sub, fused=False)
    assert_close(lc.cpu().numpy(), pc[:256].cpu().numpy(), what="layered vs wide cat")
    assert_close(la.cpu().numpy(), pa[:256].cpu().numpy(), what="layered vs wide an")


@pytest.mark.parametrize("D", [64, 128])
def test_wide_prepared_per_call_and_single_ion_agree_bitwise(D):
    Va, Vb, K, S, B = 30, 11, 8, 3, 70
    inp = synthetic.make_batch(B, atom_vocab_size=Va, bond_vocab_size=Vb, seed=12)
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=D, bond_dim=K, num_steps=S, seed=13, perturb=True)
    m = make_model(w, Va, Vb, D, K)
    d = to_dev(inp)
    ions = [(d["cat_atom"], d["cat_bond"], d["cat_connectivity"]), (d["an_atom"], d["an_bond"], d["an_connectivity"])]
    at, bt = m.atom_emb.embeddings, m.bond_emb.embeddings
    prep = m._prepared_weights("f32t")
    p2 = ops.encoder_fused(ions, at, bt, None, S, mode="f32t", prepared=prep)
    q2 = ops.encoder_fused(ions, at, bt, m._packed_weights(), S, mode="f32t")
    assert torch.equal(p2[0], q2[0]) and torch.equal(p2[1], q2[1])
    # one ion branch per call: the same rows, bit for bit
    for g in (0, 1):
        one = ops.encoder_fused([ions[g]], at, bt, None, S, mode="f32t", prepared=[prep[g]])
        assert torch.equal(one[0], p2[g])


def test_wide_atom_ids_outside_the_vocabulary_give_zero_rows():
    Va, Vb, D, K, S, B = 30, 11, 128, 8, 2, 20
    inp = synthetic.make_batch(B, atom_vocab_size=Va, bond_vocab_size=Vb, seed=21)
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=D, bond_dim=K, num_steps=S, seed=22, perturb=True)
    m = make_model(w, Va, Vb, D, K)
    bad = {k: v.copy() for k, v in inp.items()}
    bad["cat_atom"][3, 1] = Va + 5          # TF-GPU gather semantics: a zero row (ops.DEBUG_VALIDATE raises instead)
    pc, _ = m.encode_pooled(to_dev(bad), fused=True)
    wz = {k: v.copy() for k, v in w.items()}
    ext = np.zeros((Va + 6, D), np.float32)
    ext[:Va] = w["atom_embedding"]
    wz["atom_embedding"] = ext
    rc = O.encode(wz, "cat", bad["cat_atom"], bad["cat_bond"], bad["cat_connectivity"], pooled_only=True)
    assert_close(pc.cpu().numpy(), rc, what="cat pooled with an out-of-vocabulary id")


def test_wide_pipelined_plan_run_matches_single_call(wide_full):
    inp, w, m, d, pc, pa = wide_full
    plan = m.plan_batch(d)
    c, a = m.encode_pooled(d, plan=plan)
    assert torch.equal(c, pc) and torch.equal(a, pa)


def test_wide_encoder_vs_golden():
    """tests/golden/wide_d64_b5.npz (atom_dim 64, 2 steps; frozen oracle vectors): the wide encoder's pooled states and
    log_eta, and the layer-at-a-time kernels' per-layer tensors of the same model."""
    _, inp, w, outs = load_case("wide_d64_b5")
    m = MM.build_model(17, 6, atom_dim=64, bond_dim=8, fp_size=16, mixing_size=10, num_steps=2, device=DEV)
    m.load_weights(w)
    assert m.resolve_encoder_mode(14, 28) == "f32t"
    pc, pa = m.encode_pooled(to_dev(inp), fused=True)
    assert_close(pc.cpu().numpy(), outs["cat/pooled"], what="cat pooled")
    assert_close(pa.cpu().numpy(), outs["an/pooled"], what="an pooled")
    assert_close(m(inp, fused=True).cpu().numpy(), outs["final"], what="log_eta")
    tr = {}
    yl = m(inp, fused=False, trace=tr).cpu().numpy()
    assert_close(yl, outs["final"], what="log_eta layered")
    for k in ("cat/m0", "cat/agg1", "an/h2", "an/pooled", "cat/fp", "mixed"):
        assert_close(tr[k].cpu().numpy(), outs[k], what=k)


def test_wide_predict_in_chunks_equals_predict_at_once():
    """model.predict(x, batch_size) at atom_dim 128: consecutive chunks on two streams, each with its own wide workspace;
    the head runs in impnn_model_head (pooled width up to 128), so a molecule's prediction does not depend on the
    batch it sits in - bit for bit."""
    Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=128, num_steps=2, seed=5, perturb=True)
    m = MM.build_model(Va, Vb, atom_dim=128, num_steps=2, device=DEV)
    m.load_weights(w)
    inp = synthetic.make_batch(203, seed=11)
    whole = m.predict(inp)
    assert_close(whole, O.viscosity_forward(w, inp), what="log_eta")
    for bs in (32, 100):
        parts = m.predict(inp, batch_size=bs)
        assert parts.shape == whole.shape
        np.testing.assert_array_equal(parts, whole)


def _random_dense_case(rng):
    """Arbitrary multigraphs (self loops, edges naming padding atoms, id-0 holes), random shapes inside the wide
    encoder's limits; sizes kept where the (B,E,D,D) oracle finishes in seconds."""
    D = int(rng.choice([64, 128]))
    N = int(rng.integers(1, 70))
    E = min(int(rng.integers(0, 4 * N + 1)), 160)
    K = int(rng.integers(1, 9))
    S = int(rng.integers(0, 4))
    B = int(rng.integers(1, 60 if D == 128 else 120))
    Va, Vb = int(rng.integers(2, 300)), int(rng.integers(1, 400))
    ids = rng.integers(0, Va, size=(2, B, N)).astype(np.int32)
    ids[rng.random(size=ids.shape) < 0.2] = 0
    conn = rng.integers(0, N, size=(2, B, E, 2)).astype(np.int32)
    bond = rng.integers(0, Vb, size=(2, B, E)).astype(np.int32)
    inp = {"cat_atom": ids[0], "cat_bond": bond[0], "cat_connectivity": conn[0],
           "an_atom": ids[1], "an_bond": bond[1], "an_connectivity": conn[1]}
    return D, N, E, K, S, B, Va, Vb, inp


@pytest.mark.parametrize("seed", range(16))
def test_wide_encoder_fuzz_against_the_oracle(seed):
    rng = np.random.default_rng(7000 + seed)
    D, N, E, K, S, B, Va, Vb, inp = _random_dense_case(rng)
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=D, bond_dim=K, num_steps=S, seed=seed, perturb=True)
    m = make_model(w, Va, Vb, D, K)
    assert m.resolve_encoder_mode(N, E) == "f32t"
    pc, pa = m.encode_pooled(to_dev(inp), fused=True)
    rc, ra = oracle_pooled(w, inp)
    what = f"(D={D} N={N} E={E} K={K} S={S} B={B} Va={Va} Vb={Vb})"
    assert_close(pc.cpu().numpy(), rc, what="cat pooled " + what)
    assert_close(pa.cpu().numpy(), ra, what="an pooled " + what)


def test_wide_f32x3_large_batch_kernels_at_atom_dim_64():
    """atom_dim 64 at a batch that fills the chip (the 128-row GatedUpdate tiles and their 16-row pieces of the last
    round; messages stay on the exact-f32 kernel at this width): against the fp64 oracle on a sample, against the
    exact-f32 mode everywhere, shards bitwise (the halves take the 64-row kernel)."""
    Va, Vb, B = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, 1024
    inp = synthetic.make_batch(B, seed=61)
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=64, bond_dim=8, num_steps=3, seed=62, perturb=True)
    idx = np.random.default_rng(5).choice(B, size=16, replace=False)
    ref = np.concatenate(oracle_pooled(w, {k: v[idx] for k, v in inp.items()}))
    d, out = to_dev(inp), {}
    for mode in WIDE_MODES:
        m = make_model(w, Va, Vb, 64, mode=mode)
        pc, pa = m.encode_pooled(d, fused=True)
        out[mode] = torch.cat([pc, pa]).double().cpu().numpy()
        got = np.concatenate([pc.cpu().numpy()[idx], pa.cpu().numpy()[idx]]).astype(np.float64)
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max(), mode
        if mode == "f32x3":
            h = B // 2 + 3
            c0, a0 = m.encode_pooled({k: v[:h].contiguous() for k, v in d.items()}, fused=True)
            c1, a1 = m.encode_pooled({k: v[h:].contiguous() for k, v in d.items()}, fused=True)
            assert torch.equal(torch.cat([c0, c1]), pc) and torch.equal(torch.cat([a0, a1]), pa)
    assert np.abs(out["f32x3"] - out["f32t"]).max() <= 2e-5 * np.abs(out["f32t"]).max()


def test_wide_f32x3_error_within_twice_f32t_and_bitwise_properties():
    """Mode f32x3 at atom_dim 128, 6 steps (BASELINE configs[4]'s forward shape, batch 1024): error against the fp64
    oracle within twice the exact-f32 mode's (max, rms, elementwise), run-to-run and shard-concatenation bitwise."""
    Va, Vb, B = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, 1024
    inp = synthetic.make_batch(B, seed=51)
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=128, bond_dim=8, num_steps=6, seed=52, perturb=True)
    idx = np.random.default_rng(4).choice(B, size=24, replace=False)
    sub = {k: v[idx] for k, v in inp.items()}
    ref = np.concatenate(oracle_pooled(w, sub))
    d, err = to_dev(inp), {}
    for mode in WIDE_MODES:
        m = make_model(w, Va, Vb, 128, mode=mode)
        pc, pa = m.encode_pooled(d, fused=True)
        got = np.concatenate([pc.cpu().numpy()[idx], pa.cpu().numpy()[idx]]).astype(np.float64)
        dd = np.abs(got - ref)
        err[mode] = (float(dd.max() / np.abs(ref).max()), float(np.sqrt(np.mean(dd * dd)) / np.sqrt(np.mean(ref * ref))),
                     float(np.max(dd / np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max()))))
        if mode == "f32x3":
            c2, a2 = m.encode_pooled(d, fused=True)
            assert torch.equal(c2, pc) and torch.equal(a2, pa)
            h = B // 2 + 5
            c0, a0 = m.encode_pooled({k: v[:h].contiguous() for k, v in d.items()}, fused=True)
            c1, a1 = m.encode_pooled({k: v[h:].contiguous() for k, v in d.items()}, fused=True)
            assert torch.equal(torch.cat([c0, c1]), pc) and torch.equal(torch.cat([a0, a1]), pa)
    for i in range(3):
        assert err["f32x3"][i] <= 2.0 * err["f32t"][i] + 1e-7, err
    assert err["f32x3"][0] <= 1e-5 and err["f32t"][0] <= 1e-5, err


def _wide_mode_outputs(w, inp, D):
    Va, Vb = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB
    d, out = to_dev(inp), {}
    for mode in WIDE_MODES:
        pc, pa = make_model(w, Va, Vb, D, mode=mode).encode_pooled(d, fused=True)
        out[mode] = torch.cat([pc, pa]).double().cpu().numpy()
    return out


@pytest.mark.parametrize("what", ["embeddings", "bond_transform", "gate_kernels"])
def test_wide_f32x3_magnitude_sweep_at_a_chip_filling_batch(what):
    """The ruling's magnitude condition for the wide bf16x9 kernels (128-row GatedUpdate tiles, 16-row pieces, per-type
    message GEMMs with register-resident matrix planes: batch 1024 at atom_dim 128): embeddings, message weights or gate
    kernels scaled by 1e-12 ... 1e+12 - error against the fp64 oracle (a sample of molecules) within twice exact f32's
    wherever exact f32 itself is within 1e-5, the same finiteness pattern everywhere."""
    Va, Vb, B = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, 1024
    inp = synthetic.make_batch(B, seed=91)
    w0 = weights.init_weights("viscosity", Va, Vb, atom_dim=128, bond_dim=8, num_steps=2, seed=92, perturb=True)
    idx = np.random.default_rng(6).choice(B, size=10, replace=False)
    sub = {k: v[idx] for k, v in inp.items()}
    sel = np.concatenate([idx, B + idx])
    for sc in (1e-12, 1e-4, 1.0, 1e4, 1e12):
        w = dict(w0)
        for k in w0:
            hit = (what == "embeddings" and k in ("atom_embedding", "bond_embedding")) or \
                  (what == "bond_transform" and k.endswith("bond_transform")) or \
                  (what == "gate_kernels" and "/dense_" in k and k.endswith("kernel") and "_gu_" in k)
            if hit:
                w[k] = (w0[k].astype(np.float64) * sc).astype(np.float32)
        ref = np.concatenate(oracle_pooled(w, sub))
        out = _wide_mode_outputs(w, inp, 128)
        assert np.isfinite(ref).all(), (what, sc)
        assert np.array_equal(np.isfinite(out["f32t"]), np.isfinite(out["f32x3"])), (what, sc)
        if not np.isfinite(out["f32t"]).all():
            continue
        scale = np.abs(ref).max()
        err = {m: float(np.abs(out[m][sel] - ref).max() / scale) for m in WIDE_MODES}
        if err["f32t"] <= 1e-5:
            assert err["f32x3"] <= 2.0 * err["f32t"] + 1e-7 and err["f32x3"] <= 1e-5, (what, sc, err)
        else:  # ill-conditioned corner (saturated gates amplify any f32 rounding): the same order of magnitude
            assert err["f32x3"] <= 4.0 * err["f32t"] + 1e-7, (what, sc, err)


def test_wide_f32x3_propagates_nan_like_f32t():
    """A NaN in an embedding row or a weight makes the same molecules' outputs NaN in both wide modes and leaves every
    other molecule's output as it was; an infinity never yields a finite value where exact f32 reports none."""
    Va, Vb, B = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, 1024
    inp = synthetic.make_batch(B, seed=93)
    w0 = weights.init_weights("viscosity", Va, Vb, atom_dim=128, bond_dim=8, num_steps=2, seed=94, perturb=True)
    clean = _wide_mode_outputs(w0, inp, 128)
    for bad in (np.nan, np.inf):
        for key, where in (("atom_embedding", (17, 5)), ("bond_embedding", (9, 2)), ("cat_gu_1/dense_r/kernel", (40, 3)),
                           ("an_bmm_0/bond_transform", (3, 7, 11))):
            w = {k: v.copy() for k, v in w0.items()}
            w[key][where] = bad
            res = _wide_mode_outputs(w, inp, 128)
            fin_t, fin_x = np.isfinite(res["f32t"]).all(axis=1), np.isfinite(res["f32x3"]).all(axis=1)
            assert not (fin_x & ~fin_t).any(), (bad, key)   # never finite where exact f32 is not
            if np.isnan(bad):
                assert not fin_t.all(), (bad, key)          # the poison reached some molecule
                assert np.array_equal(fin_t, fin_x), (bad, key, int(fin_t.sum()), int(fin_x.sum()))
            # (an infinite gate kernel saturates the exact-f32 sigmoid to 0 / 1 - finite everywhere - while its three-term
            #  split is (inf, nan, nan): the emulation is the more conservative of the two, encoder_typed.hip)
            both = fin_t & fin_x
            untouched = both & (np.abs(res["f32t"] - clean["f32t"]).max(axis=1) == 0)
            assert np.array_equal(res["f32x3"][untouched], clean["f32x3"][untouched]), (bad, key)


@pytest.mark.parametrize("mode", WIDE_MODES)
def test_wide_encoder_single_atom_anions(mode):
    """Halide-like anions (one atom, no bond) beside ordinary cations at a chip-filling batch: the anion's rows are a
    sliver of the row space (partial tiles, 16-row pieces), every anion with the same atom id must come out the same."""
    Va, Vb, B = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, 2048
    inp = synthetic.make_batch(B, seed=15)
    inp["an_atom"][:, 1:] = 0
    inp["an_bond"][:] = 0
    inp["an_connectivity"][:] = 0
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=128, bond_dim=8, num_steps=2, seed=16, perturb=True)
    idx = np.concatenate([np.arange(4), np.random.default_rng(2).choice(B, size=8, replace=False), np.arange(B - 4, B)])
    rc, ra = oracle_pooled(w, {k: v[idx] for k, v in inp.items()})
    pc, pa = make_model(w, Va, Vb, 128, mode=mode).encode_pooled(to_dev(inp), fused=True)
    assert_close(pa.cpu().numpy()[idx], ra, what="an pooled")
    assert_close(pc.cpu().numpy()[idx], rc, what="cat pooled")
    ids = inp["an_atom"][:, 0]
    first = {int(v): int(np.argmax(ids == v)) for v in np.unique(ids)}
    ref_rows = pa[torch.as_tensor([first[int(v)] for v in ids], device=pa.device)]
    assert torch.equal(pa, ref_rows)


@pytest.mark.parametrize("mode", WIDE_MODES)
@pytest.mark.parametrize("D,S,B", [(128, 6, 1024), (64, 3, 512)])
def test_wide_encoder_explicit_hydrogen_shape(D, S, B, mode):
    """The padded shape of the reference's real (explicit-hydrogen) data sets at wide states - N = 160, E = 640
    (src/featurize.py:45, train_viscosity.py:95,288-289 with atom_dim=128, num_steps=6): the wide encoder takes E <= 1024
    (round 2: 512, so this shape ran layer at a time).  Sampled molecules vs the oracle, shard-concat bitwise, and the
    layer-at-a-time HIP path."""
    Va, Vb, N, E = synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, 160, 640
    inp = synthetic.make_explicit_h_batch(B, max_atoms=N, max_edges=E, seed=51)
    valid = (inp["cat_connectivity"][:, :, 0] > 0) & (inp["cat_connectivity"][:, :, 1] > 0)
    assert valid.sum(axis=1).max() > 512
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=D, bond_dim=8, num_steps=S, seed=52, perturb=True)
    m = make_model(w, Va, Vb, D, mode=mode)
    assert m.resolve_encoder_mode(N, E) == mode and m.fused_supported(N, E)
    d = to_dev(inp)
    pc, pa = m.encode_pooled(d, fused=True)
    torch.cuda.synchronize()
    assert torch.isfinite(pc).all() and torch.isfinite(pa).all()
    idx = np.random.default_rng(8).choice(B, size=10, replace=False)
    idx[0] = int(valid.sum(axis=1).argmax())
    rc, ra = oracle_pooled(w, {k: v[idx] for k, v in inp.items()})
    assert_close(pc.cpu().numpy()[idx], rc, what="explicit-H cat pooled (sample)")
    assert_close(pa.cpu().numpy()[idx], ra, what="explicit-H an pooled (sample)")
    h = B // 2 + 9
    c0, a0 = m.encode_pooled({k: v[:h].contiguous() for k, v in d.items()}, fused=True)
    c1, a1 = m.encode_pooled({k: v[h:].contiguous() for k, v in d.items()}, fused=True)
    assert torch.equal(torch.cat([c0, c1]), pc) and torch.equal(torch.cat([a0, a1]), pa)
    lc, la = m.encode_pooled({k: v[:32].contiguous() for k, v in d.items()}, fused=False)
    assert_close(lc.cpu().numpy(), pc[:32].cpu().numpy(), what="explicit-H layered vs wide cat")
    assert_close(la.cpu().numpy(), pa[:32].cpu().numpy(), what="explicit-H layered vs wide an")
