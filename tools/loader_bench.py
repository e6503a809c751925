"""Batch preparation: host list handling (the reference's way, numpy-restated) + H2D copy
against impnn_batch_assemble on a resident dataset.  python tools/loader_bench.py [--batch 4096]"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ionic_mpnn_amd import data, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--records", type=int, default=20000)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    recs, vocab = synthetic.make_id_records(a.records, seed=0, min_atoms=8, max_atoms=40, atom_vocab=123, bond_vocab=71)
    host = data.IonPairDataset(recs, vocab)
    res = data.ResidentIonPairDataset(recs, vocab)
    rng = np.random.default_rng(0)
    idx = rng.permutation(a.records)[:a.batch]
    t0 = time.perf_counter()
    for _ in range(3):
        hb = host.build_inputs(idx.tolist())
        db = {k: torch.from_numpy(v).cuda() for k, v in hb.items()}
    torch.cuda.synchronize()
    host_ms = (time.perf_counter() - t0) / 3 * 1e3
    didx = torch.from_numpy(idx.astype(np.int32)).cuda()
    for _ in range(5):
        gb = res.build_inputs(didx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters):
        gb = res.build_inputs(didx)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / a.iters * 1e3
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    from ionic_mpnn_amd import ops
    L = 2 * res.max_edges
    ev0.record()
    for _ in range(a.iters):
        ops.batch_assemble(didx, res.ions, res.max_atoms, L, t_flat=res.temperature)
    ev1.record()
    torch.cuda.synchronize()
    bytes_out = sum(v.numel() * v.element_size() for v in gb.values())
    for k in hb:
        assert np.array_equal(gb[k].cpu().numpy(), hb[k]), k
    dev_ms = ev0.elapsed_time(ev1) / a.iters
    print(json.dumps({"batch": a.batch, "N": res.max_atoms, "slots": L, "host_build_plus_h2d_ms": host_ms,
                      "gpu_assemble_call_ms": gpu_ms, "gpu_assemble_stream_ms": dev_ms, "bytes_written": bytes_out,
                      "write_GBs": bytes_out / (dev_ms * 1e-3) / 1e9, "speedup_vs_host": host_ms / gpu_ms}))


if __name__ == "__main__":
    main()
