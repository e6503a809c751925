"""Generates tests/golden/plumbing_golden.json by IMPORTING the reference's own
utils/mp_utils.py (numpy + matplotlib only) in the build container and recording its outputs on
fixed inputs.  The reference never travels to the GPU box; this JSON (data only) does.

    python tests/golden/make_plumbing_golden.py [/root/reference]
"""
import importlib.util
import json
import sys
from pathlib import Path

import numpy as np

ref_root = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
spec = importlib.util.spec_from_file_location("ref_mp_utils", ref_root / "utils" / "mp_utils.py")
import matplotlib
matplotlib.use("Agg")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

rng = np.random.default_rng(20240601)
cases = {"pad_sequences_1d": [], "preprocess_edges_and_bonds": [], "r2_numpy": []}

# hand cases from SURVEY.md 8c
cases["pad_sequences_1d"].append({"seqs": [[3, 4, 2], [9]], "max_len": 4, "pad_val": 0})
cases["pad_sequences_1d"].append({"seqs": [[1], [2, 3], []], "max_len": 2, "pad_val": 0})
for _ in range(4):
    n = int(rng.integers(1, 6))
    L = int(rng.integers(3, 9))
    seqs = [[int(x) for x in rng.integers(1, 50, size=int(rng.integers(0, L + 1)))] for _ in range(n)]
    cases["pad_sequences_1d"].append({"seqs": seqs, "max_len": L, "pad_val": 0})
for c in cases["pad_sequences_1d"]:
    out = ref.pad_sequences_1d(c["seqs"], c["max_len"], c["pad_val"])
    c["out"] = out.tolist()
    c["dtype"] = str(out.dtype)

chain = {"edges": [[(0, 1), (1, 0), (1, 2), (2, 1)]], "bonds": [[5, 5, 7, 7]], "max_edges": 4}
cases["preprocess_edges_and_bonds"].append(chain)
# truncation: len(e2) >= 2*max_edges
cases["preprocess_edges_and_bonds"].append(
    {"edges": [[(0, 1), (1, 0), (1, 2), (2, 1), (2, 3), (3, 2)]], "bonds": [[1, 1, 2, 2, 3, 3]], "max_edges": 4})
cases["preprocess_edges_and_bonds"].append(
    {"edges": [[(0, 1), (1, 0)], [], [(2, 0), (0, 2), (1, 2), (2, 1)]], "bonds": [[4, 4], [], [9, 9, 1, 1]],
     "max_edges": 4})
for _ in range(4):
    m = int(rng.integers(1, 5))
    max_e = int(rng.integers(2, 9))
    edges, bonds = [], []
    for _ in range(m):
        ne = int(rng.integers(0, max_e + 3))  # sometimes longer than max_edges -> truncation
        e = [(int(a), int(b)) for a, b in rng.integers(0, 7, size=(ne, 2))]
        edges.append(e)
        bonds.append([int(x) for x in rng.integers(1, 12, size=ne)])
    cases["preprocess_edges_and_bonds"].append({"edges": edges, "bonds": bonds, "max_edges": max_e})
for c in cases["preprocess_edges_and_bonds"]:
    e, b = ref.preprocess_edges_and_bonds(c["edges"], c["bonds"], c["max_edges"])
    c["edges"] = [[list(p) for p in mol] for mol in c["edges"]]
    c["out_edges"], c["out_bonds"] = e.tolist(), b.tolist()
    c["out_edges_shape"], c["out_bonds_shape"] = list(e.shape), list(b.shape)
    c["dtype"] = str(e.dtype)

cases["r2_numpy"].append({"y_true": [1.0, 2.0, 3.0], "y_pred": [1.1, 1.9, 3.2]})
for _ in range(3):
    n = int(rng.integers(3, 20))
    yt = rng.normal(size=n)
    cases["r2_numpy"].append({"y_true": yt.tolist(), "y_pred": (yt + 0.3 * rng.normal(size=n)).tolist()})
for c in cases["r2_numpy"]:
    c["out"] = float(ref.r2_numpy(np.array(c["y_true"]), np.array(c["y_pred"])))

out_path = Path(__file__).resolve().parent / "plumbing_golden.json"
out_path.write_text(json.dumps(cases, indent=1))
print("wrote", out_path)
