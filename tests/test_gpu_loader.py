"""f2: batch assembly on the GPU must reproduce, bit for bit, the reference's host list
handling (train_viscosity.py:248-314, utils/mp_utils.py:12-45) as restated - and pinned against the
reference's own helpers - in ionic_mpnn_amd.data.IonPairDataset."""
import numpy as np
import pytest
import torch

from ionic_mpnn_amd import data, synthetic

pytestmark = pytest.mark.gpu


def _same(dev_inputs, host_inputs):
    assert set(dev_inputs) == set(host_inputs)
    for k, v in host_inputs.items():
        got = dev_inputs[k].cpu().numpy()
        assert got.dtype == v.dtype and got.shape == v.shape, k
        assert np.array_equal(got, v), k


@pytest.mark.parametrize("kind", ["viscosity", "melting_point"])
def test_batches_match_host_loader(kind):
    recs, vocab = synthetic.make_id_records(300, seed=3, min_atoms=1, max_atoms=40, atom_vocab=120, bond_vocab=70,
                                            kind=kind)
    host = data.IonPairDataset(recs, vocab)
    res = data.ResidentIonPairDataset(recs, vocab)
    assert (res.max_atoms, res.max_edges) == (host.max_atoms, host.max_edges)
    rng = np.random.default_rng(0)
    for idx in (list(range(300)), rng.permutation(300)[:77].tolist(), [5], [7, 7, 7, -1], []):
        _same(res.build_inputs(idx), host.build_inputs([i % 300 for i in idx]))
    dev_idx = torch.tensor([4, 250, 0], dtype=torch.int32, device="cuda")
    _same(res.build_inputs(dev_idx), host.build_inputs([4, 250, 0]))


def test_ragged_edge_cases_and_truncation():
    """Empty molecules, edge/bond lists of different length (zip semantics), and truncation at 2*max_edges."""
    recs, vocab = synthetic.make_id_records(40, seed=9, min_atoms=2, max_atoms=9, kind="viscosity")
    recs[3]["cation"] = {"atom_ids": [], "bond_ids": [], "edge_indices": [], "num_atoms": 0}
    recs[4]["anion"]["bond_ids"] = recs[4]["anion"]["bond_ids"][:-3]       # fewer bond ids than edges
    recs[5]["cation"]["edge_indices"] = recs[5]["cation"]["edge_indices"][:1]  # fewer edges than bond ids
    host = data.IonPairDataset(recs, vocab)
    res = data.ResidentIonPairDataset(recs, vocab)
    idx = list(range(40))
    _same(res.build_inputs(idx), host.build_inputs(idx))
    # a caller-chosen max_edges below the longest list truncates exactly like utils/mp_utils.py:40-41
    short = 3
    got = res.build_inputs(idx, max_edges=short)
    for key, ion in (("cat", "cat"), ("an", "an")):
        conn, bond = data.preprocess_edges_and_bonds([host.ions[ion]["edges"][i] for i in idx],
                                                     [host.ions[ion]["bonds"][i] for i in idx], short)
        assert np.array_equal(got[f"{key}_connectivity"].cpu().numpy(), conn)
        assert np.array_equal(got[f"{key}_bond"].cpu().numpy(), bond)
    with pytest.raises(ValueError):
        res.build_inputs(idx, max_atoms=host.max_atoms - 1)
    with pytest.raises(IndexError):
        res.build_inputs([40])


def test_assembled_batch_feeds_the_encoder():
    """records -> resident dataset -> batch on the GPU -> fused encoder == the same through the host loader."""
    from ionic_mpnn_amd import model as M, weights
    recs, vocab = synthetic.make_id_records(64, seed=1, min_atoms=3, max_atoms=20, atom_vocab=30, bond_vocab=8)
    host = data.IonPairDataset(recs, vocab)
    res = data.ResidentIonPairDataset(recs, vocab)
    w = weights.init_weights("viscosity", host.atom_vocab_size, host.bond_vocab_size, seed=2, perturb=True)
    m = M.build_model(host.atom_vocab_size, host.bond_vocab_size, num_steps=weights.num_steps_of(w), device="cuda")
    m.load_weights(w)
    idx = list(range(64))
    a = m.predict(res.build_inputs(idx))
    b = m.predict(host.build_inputs(idx))
    assert a.shape == (64, 1) and np.array_equal(a, b)
