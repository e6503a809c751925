#!/usr/bin/env python3
"""One-off extension of the committed fuzz tests (tests/test_gpu_encoder.py, tests/test_gpu_wide.py): the same random
dense multigraph cases for further seeds, every exact-f32 form the shape allows, against the fp64 oracle.
python tools/extended_fuzz.py [--first 24] [--count 150]      (GPU box; test infrastructure, uses oracle/)"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import test_gpu_encoder as TE  # noqa: E402
import test_gpu_wide as TW  # noqa: E402
from conftest import assert_close  # noqa: E402
from ionic_mpnn_amd import weights  # noqa: E402
from oracle import mpnn_oracle as O  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--first", type=int, default=24)
ap.add_argument("--count", type=int, default=150)
args = ap.parse_args()
bad, ran = [], 0
for seed in range(args.first, args.first + args.count):
    rng = np.random.default_rng(1000 + seed)
    N, E, K, S, B, Va, Vb, inp = TE._random_dense_case(rng)
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=32, bond_dim=K, num_steps=S, seed=seed, perturb=True)
    rc = O.encode(w, "cat", inp["cat_atom"], inp["cat_bond"], inp["cat_connectivity"], pooled_only=True)
    ra = O.encode(w, "an", inp["an_atom"], inp["an_bond"], inp["an_connectivity"], pooled_only=True)
    for mode in ("f32t", "f32x3", "f32"):
        m = TE.make_model(w, Va, Vb, K=K, mode=mode)
        if m.resolve_encoder_mode(N, E) != mode:
            continue
        try:
            pc, pa = m.encode_pooled(TE.to_dev(inp), fused=True)
            assert_close(pc.cpu().numpy(), rc, what="cat")
            assert_close(pa.cpu().numpy(), ra, what="an")
            ran += 1
        except Exception as e:  # noqa: BLE001
            bad.append(("typed", seed, mode, (N, E, K, S, B, Va, Vb), str(e)[:200]))
    rng = np.random.default_rng(7000 + seed)
    D, N, E, K, S, B, Va, Vb, inp = TW._random_dense_case(rng)
    w = weights.init_weights("viscosity", Va, Vb, atom_dim=D, bond_dim=K, num_steps=S, seed=seed, perturb=True)
    rc, ra = TW.oracle_pooled(w, inp)
    for mode in TW.WIDE_MODES:
        try:
            pc, pa = TW.make_model(w, Va, Vb, D, K, mode=mode).encode_pooled(TW.to_dev(inp), fused=True)
            assert_close(pc.cpu().numpy(), rc, what="cat")
            assert_close(pa.cpu().numpy(), ra, what="an")
            ran += 1
        except Exception as e:  # noqa: BLE001
            bad.append(("wide", seed, mode, (D, N, E, K, S, B, Va, Vb), str(e)[:200]))
    if seed % 25 == 0:
        print(f"seed {seed}: {ran} runs, {len(bad)} failures", flush=True)
print(f"{ran} encoder runs over seeds {args.first} .. {args.first + args.count - 1}: {len(bad)} failures")
for b in bad:
    print(b)
sys.exit(1 if bad else 0)
