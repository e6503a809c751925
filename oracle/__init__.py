"""CPU oracle for the ionic-mpnn message-passing forward path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``ionic_mpnn_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.

Parity status: the layer arithmetic is **parity unpinned** (the reference holds
no tests, fixtures or weights for it and TensorFlow/Keras is not installable
here); the input plumbing (`pad_sequences_1d`,
`preprocess_edges_and_bonds`, `r2_numpy`) **is pinned** against the reference's
own importable ``utils/mp_utils.py`` (tests/golden/plumbing_golden.json).
"""
