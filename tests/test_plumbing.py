"""CPU: host-side input plumbing against fixtures recorded from the reference's own
utils/mp_utils.py (tests/golden/make_plumbing_golden.py) - this part of the oracle IS pinned."""
import json
import pickle

import numpy as np
import pytest

from conftest import GOLDEN
from ionic_mpnn_amd import data, synthetic
from oracle import mpnn_oracle as O

G = json.loads((GOLDEN / "plumbing_golden.json").read_text())


@pytest.mark.parametrize("impl", [data, O], ids=["product", "oracle"])
def test_pad_sequences_1d(impl):
    for c in G["pad_sequences_1d"]:
        out = impl.pad_sequences_1d(c["seqs"], c["max_len"], c["pad_val"])
        assert str(out.dtype) == c["dtype"] == "int32"
        np.testing.assert_array_equal(out, np.array(c["out"], dtype=np.int32).reshape(len(c["seqs"]), c["max_len"]))


@pytest.mark.parametrize("impl", [data, O], ids=["product", "oracle"])
def test_preprocess_edges_and_bonds(impl):
    for c in G["preprocess_edges_and_bonds"]:
        edges = [[tuple(p) for p in mol] for mol in c["edges"]]
        e, b = impl.preprocess_edges_and_bonds(edges, c["bonds"], c["max_edges"])
        assert e.dtype == np.int32 and b.dtype == np.int32
        assert list(e.shape) == c["out_edges_shape"] and list(b.shape) == c["out_bonds_shape"]
        np.testing.assert_array_equal(e, np.array(c["out_edges"], dtype=np.int32))
        np.testing.assert_array_equal(b, np.array(c["out_bonds"], dtype=np.int32))


def test_chain_fact_from_survey():
    # 3-atom chain -> every bond 4x
    e, b = data.preprocess_edges_and_bonds([[(0, 1), (1, 0), (1, 2), (2, 1)]], [[5, 5, 7, 7]], 4)
    assert e[0].tolist() == [[0, 1], [1, 0], [1, 0], [0, 1], [1, 2], [2, 1], [2, 1], [1, 2]]
    assert b[0].tolist() == [5, 5, 5, 5, 7, 7, 7, 7]


@pytest.mark.parametrize("impl", [data, O], ids=["product", "oracle"])
def test_r2_numpy(impl):
    for c in G["r2_numpy"]:
        assert impl.r2_numpy(np.array(c["y_true"]), np.array(c["y_pred"])) == pytest.approx(c["out"], rel=1e-12)


def test_pad_too_long_is_an_error():
    with pytest.raises(ValueError):
        data.pad_sequences_1d([[1, 2, 3]], 2)


def test_dataset_restates_trainer_prep(tmp_path):
    """config 1 plumbing: synthetic pkl in the src/dataset.py:51-62 schema -> loader -> 7 inputs,
    checked against a literal restatement of train_viscosity.py:248-314."""
    recs, vocab = synthetic.make_id_records(40, seed=3)
    dp, vp = tmp_path / "viscosity_id_data.pkl", tmp_path / "vocab.pkl"
    dp.write_bytes(pickle.dumps(recs))
    vp.write_bytes(pickle.dumps(vocab))
    recs2, vocab2 = data.load_id_dataset(dp, vp)
    ds = data.IonPairDataset(recs2, vocab2)
    assert ds.atom_vocab_size == vocab["atom_vocab_size"] + 1 and ds.bond_vocab_size == vocab["bond_vocab_size"] + 1
    idx = list(range(32))
    x = ds.build_inputs(idx)
    # literal restatement with the oracle's list-based helpers
    max_atoms = max(len(r[ion]["atom_ids"]) for r in recs for ion in ("cation", "anion"))
    max_edges = max(len(r[ion]["edge_indices"]) for r in recs for ion in ("cation", "anion"))
    assert (ds.max_atoms, ds.max_edges) == (max_atoms, max_edges)
    for key, ion in (("cat", "cation"), ("an", "anion")):
        atoms = [[a + 1 for a in recs[i][ion]["atom_ids"]] for i in idx]
        bonds = [[b + 1 for b in recs[i][ion]["bond_ids"]] for i in idx]
        edges = [recs[i][ion]["edge_indices"] for i in idx]
        e, b = O.preprocess_edges_and_bonds(edges, bonds, max_edges)
        np.testing.assert_array_equal(x[f"{key}_atom"], O.pad_sequences_1d(atoms, max_atoms))
        np.testing.assert_array_equal(x[f"{key}_connectivity"], e)
        np.testing.assert_array_equal(x[f"{key}_bond"], b)
        assert x[f"{key}_connectivity"].shape == (32, 2 * max_edges, 2)
    assert x["temperature"].shape == (32, 1) and x["temperature"].dtype == np.float32
    # edge indices are NOT shifted: atom 0 appears as an endpoint
    assert (x["cat_connectivity"][..., 0] == 0).any()


def test_shard_bounds_cover_batch_exactly():
    for n in (0, 1, 7, 4096, 65537):
        for w in (1, 2, 3, 8):
            spans = [data.shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
