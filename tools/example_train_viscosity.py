"""The reference's trainer flow (train_viscosity.py main(), :236-372) on this package, end to end on the GPU:
id records -> HBM-resident dataset (impnn_batch_assemble) -> build_model -> fit (hipGraph-replayed steps, Adam with
clipnorm, early stopping) -> predict (fused encoder) -> R2 / MAE.  The real viscosity_id_data.pkl is not part of the
reference repository, so the records are synthetic (ionic_mpnn_amd.synthetic.make_id_records) with a target that
is a known function of the graphs - the point is the plumbing, not chemistry.
    python tools/example_train_viscosity.py [--records 600] [--epochs 30]"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ionic_mpnn_amd import build_model, data, synthetic  # noqa: E402
from ionic_mpnn_amd.train import Adam, EarlyStopping  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=600)
    ap.add_argument("--epochs", type=int, default=30)
    ap.add_argument("--seed", type=int, default=42)
    a = ap.parse_args()
    recs, vocab = synthetic.make_id_records(a.records, seed=a.seed, min_atoms=4, max_atoms=24, atom_vocab=30, bond_vocab=6)
    for r in recs:  # a learnable target: depends on composition, size and temperature
        n_c, n_a = r["cation"]["num_atoms"], r["anion"]["num_atoms"]
        r["log_eta"] = 0.08 * n_c + 0.05 * n_a + 0.02 * sum(r["cation"]["atom_ids"]) / n_c + 300.0 / r["T"]
    ds = data.ResidentIonPairDataset(recs, vocab)                       # train_viscosity.py:248-289
    y = np.asarray(ds.log_eta, np.float32)
    rng = np.random.RandomState(a.seed)                                 # :269-286 (80/10/10 split)
    idx = rng.permutation(len(ds))
    n_tr, n_dev = int(0.8 * len(ds)), int(0.1 * len(ds))
    parts = {"train": idx[:n_tr], "dev": idx[n_tr:n_tr + n_dev], "test": idx[n_tr + n_dev:]}
    x = {k: ds.build_inputs(v.tolist()) for k, v in parts.items()}      # :291-314, on the GPU
    model = build_model(ds.atom_vocab_size, ds.bond_vocab_size, num_steps=3)
    model.compile(Adam(1e-3, clipnorm=1.0))                             # :227-230
    t0 = time.perf_counter()
    hist = model.fit(x["train"], y[parts["train"]], validation_data=(x["dev"], y[parts["dev"]]), epochs=a.epochs,
                     batch_size=32, callbacks=[EarlyStopping(monitor="val_loss", patience=50, restore_best_weights=True)],
                     seed=a.seed)                                       # :328-338
    torch.cuda.synchronize()
    secs = time.perf_counter() - t0
    out = {"epochs_run": len(hist.history["loss"]), "seconds": secs, "first_loss": hist.history["loss"][0],
           "last_loss": hist.history["loss"][-1], "best_val_loss": min(hist.history["val_loss"])}
    for name, ids in parts.items():                                     # :361-369
        pred = model.predict(x[name]).flatten()
        out[f"{name}_r2"] = float(data.r2_numpy(y[ids], pred))
        out[f"{name}_mae"] = float(np.mean(np.abs(y[ids] - pred)))
    print(json.dumps(out))
    return out


if __name__ == "__main__":
    main()
