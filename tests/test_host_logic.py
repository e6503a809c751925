"""CPU: the Keras-layer surface (names, configs, weight shapes, build idempotence, model wiring).
Weights are created on the CPU device here; compute calls are GPU-only and tested under -m gpu."""
import numpy as np
import pytest
import torch

import ionic_mpnn_amd as M
from ionic_mpnn_amd import layers as L
from ionic_mpnn_amd import weights as W

CPU = torch.device("cpu")


def test_bond_matrix_message_surface():
    L.reset_uids()
    lyr = M.BondMatrixMessage(32, 8, name="cat_bmm_0", device=CPU)
    assert (lyr.atom_dim, lyr.bond_dim, lyr.name, lyr.trainable) == (32, 8, "cat_bmm_0", True)
    lyr.build([(None, None, 32), (None, None, 8), (None, None, 2)])
    assert tuple(lyr.bond_transform.shape) == (8, 32, 32)            # models/layers.py:94-98
    assert lyr.weight_names() == ["bond_transform"]
    lim = np.sqrt(6.0 / (2 * 8 * 32))                                 # keras glorot fans for (K,D,D)
    assert float(lyr.bond_transform.abs().max()) <= lim
    cfg = lyr.get_config()
    assert cfg["atom_dim"] == 32 and cfg["bond_dim"] == 8 and cfg["name"] == "cat_bmm_0"   # :119-125
    again = M.BondMatrixMessage.from_config({k: cfg[k] for k in ("atom_dim", "bond_dim", "name")} | {"device": CPU})
    assert again.get_config() == cfg
    w0 = lyr.bond_transform
    lyr.build(None)                                                   # idempotent
    assert lyr.bond_transform is w0


def test_gated_update_surface_and_alias():
    L.reset_uids()
    g0, g1 = M.GatedUpdate(32, device=CPU), M.GatedUpdate(32, device=CPU)
    assert (g0.name, g1.name) == ("gated_update", "gated_update_1")   # keras auto names
    assert M.GRUUpdate is M.GatedUpdate
    g0.build(None)
    assert g0.weight_names() == ["dense_z/kernel", "dense_z/bias", "dense_r/kernel", "dense_r/bias",
                                 "dense_h/kernel", "dense_h/bias", "layernorm/gamma", "layernorm/beta"]
    assert tuple(g0.dense_z.kernel.shape) == (64, 32) and tuple(g0.dense_h.bias.shape) == (32,)
    assert float(g0.gamma.min()) == 1.0 and float(g0.beta.abs().max()) == 0.0 and g0.epsilon == 1e-3
    assert g0.dropout_rate == 0.0 and g0.get_config()["atom_dim"] == 32
    assert len(g0.weights) == 8                                       # 8 variables each


def test_registry_and_parameterless_layers():
    for n in ("Reduce", "BondMatrixMessage", "GatedUpdate", "GlobalSumPool", "ComputeLogEta", "SliceParamB"):
        assert L.get_registered(n).__name__ == n
        assert L.get_registered(f"Custom>{n}").__name__ == n
    r = M.Reduce(name="cat_reduce_0", device=CPU)
    assert r.weights == [] and r.get_config()["name"] == "cat_reduce_0"
    with pytest.raises(TypeError):
        M.Reduce(bogus=1)


def test_build_model_wiring_matches_reference_names():
    L.reset_uids()
    m = M.build_model(124, 72, device=CPU)                             # defaults of train_viscosity.py:139-147
    assert (m.atom_dim, m.bond_dim, m.fp_size, m.mixing_size, m.num_steps) == (32, 8, 32, 20, 4)
    names = [l.name for l in m.layers]
    for i in range(4):
        assert f"cat_bmm_{i}" in names and f"an_reduce_{i}" in names
    # 2*S independent GatedUpdate instances; the transfer script's layer names exist (:214-220)
    assert [n for n in names if n.startswith("gated_update")] == ["gated_update"] + [f"gated_update_{i}" for i in range(1, 8)]
    assert m.get_layer("mix_cat_an") is m.mix and m.get_layer("param_B").name == "param_B"
    assert m.branches["cat"]["update"][0] is not m.branches["an"]["update"][0]
    assert m.branches["cat"]["bmm"][1].bond_transform is not m.branches["cat"]["bmm"][0].bond_transform
    sd = m.state_dict()
    ref = W.init_weights("viscosity", 124, 72, num_steps=4)
    assert set(sd) == set(ref) and all(sd[k].shape == ref[k].shape for k in ref)
    m.load_weights(ref)
    np.testing.assert_array_equal(m.state_dict()["an_gu_3/dense_h/kernel"], ref["an_gu_3/dense_h/kernel"])
    with pytest.raises(ValueError):
        bad = dict(ref); bad["atom_embedding"] = bad["atom_embedding"][:, :5]; m.load_weights(bad)
    with pytest.raises(ValueError):
        m.get_layer("nope")


def test_melting_point_model_uses_squared_bond_dim():
    m = M.build_melting_point_model(20, 6, atom_dim=8, num_steps=2, device=CPU)
    assert m.bond_dim == 64 and tuple(m.bond_emb.embeddings.shape) == (6, 64)     # train_melting_point.py:146,150
    assert tuple(m.branches["an"]["bmm"][1].bond_transform.shape) == (64, 8, 8)
    assert tuple(m.mp_out.kernel.shape) == (32, 1) and tuple(m.mp_hidden.kernel.shape) == (20, 32)


def test_initialisers():
    rng = np.random.default_rng(0)
    k = W.glorot_uniform(rng, (64, 32))
    assert k.dtype == np.float32 and abs(k).max() <= np.sqrt(6 / 96)
    e = W.embedding_uniform(rng, (124, 32))
    assert abs(e).max() <= 0.05
    w = W.init_weights("viscosity", 124, 72, num_steps=3)
    assert W.num_steps_of(w) == 3 and float(abs(w["cat_gu_0/dense_z/bias"]).max()) == 0.0


def test_flatten_ion_ragged_layout():
    """The resident layout of the loader (data.flatten_ion): raw ids, zip()-cut edge/bond lists, int32 offsets."""
    from ionic_mpnn_amd import data
    recs = [
        {"cation": {"atom_ids": [3, 1], "bond_ids": [2, 2], "edge_indices": [(0, 1), (1, 0)]}},
        {"cation": {"atom_ids": [], "bond_ids": [], "edge_indices": []}},
        {"cation": {"atom_ids": [5, 6, 7], "bond_ids": [1, 1, 4], "edge_indices": [(0, 1), (1, 0), (1, 2), (2, 1)]}},
    ]
    f = data.flatten_ion(recs, "cation")
    assert f["atom_off"].tolist() == [0, 2, 2, 5] and f["edge_off"].tolist() == [0, 2, 2, 5]
    assert f["atom_flat"][:5].tolist() == [3, 1, 5, 6, 7]
    assert f["edge_flat"][:5].tolist() == [[0, 1], [1, 0], [0, 1], [1, 0], [1, 2]]  # 4th edge cut: only 3 bond ids
    assert f["bond_flat"][:5].tolist() == [2, 2, 1, 1, 4]
    assert all(v.dtype == np.int32 for v in f.values())


def test_id_dataset_is_validated_once_on_the_host():
    """Corrupt records raise when the dataset is built (the reference's tf.gather / scatter_nd raise on them on the
    CPU); the kernels themselves treat out-of-range indices as padding (include/impnn.h)."""
    import copy
    import pytest
    from ionic_mpnn_amd import data, synthetic
    recs, vocab = synthetic.make_id_records(6, seed=1)
    data.IonPairDataset(recs, vocab)  # clean
    for mutate, what in ((lambda r: r["cation"]["atom_ids"].__setitem__(0, vocab["atom_vocab_size"]), "atom_ids"),
                         (lambda r: r["anion"]["bond_ids"].__setitem__(0, -2), "bond_ids"),
                         (lambda r: r["anion"]["edge_indices"].__setitem__(0, (0, 99)), "edge_indices")):
        bad = copy.deepcopy(recs)
        mutate(bad[2])
        with pytest.raises(ValueError, match=what):
            data.IonPairDataset(bad, vocab)
