"""Message backward entry (impnn_message_reduce_typed_bwd) timed alone: python tools/msg_bwd_bench.py [--batch 32] [--dim 128]
IMPNN_MESSAGE_BWD=valu | mb (matrix cores, balanced segment ranges) | mo (matrix cores, one workgroup per type)"""
import argparse
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ionic_mpnn_amd import _lib, ops, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--explicit-h", action="store_true", help="N = 160, E = 640 molecules (synthetic.make_explicit_h_batch)")
a = ap.parse_args()
dev = torch.device("cuda:0")
B, D, Vb = a.batch, a.dim, synthetic.DEFAULT_VB
inp = synthetic.make_explicit_h_batch(B, seed=0) if a.explicit_h else synthetic.make_batch(B, seed=0)
conn = torch.from_numpy(inp["cat_connectivity"]).to(dev)
bond = torch.from_numpy(inp["cat_bond"]).to(dev)
N, E = inp["cat_atom"].shape[1], conn.shape[1]
g = torch.Generator(device=dev).manual_seed(0)
h = torch.randn(B, N, D, device=dev, generator=g)
dagg = torch.randn(B, N, D, device=dev, generator=g)
mats = torch.randn(Vb, D, D, device=dev, generator=g) / D ** 0.5
dh, dA = torch.zeros(B, N, D, device=dev), torch.zeros_like(mats)
lib = _lib.load()
ws = torch.empty(int(lib.impnn_bmm_message_typed_bwd_workspace_bytes(B, E, Vb)), dtype=torch.uint8, device=dev)


def call(ready):
    _lib.check(lib.impnn_message_reduce_typed_bwd(ops.ptr(h), ops.ptr(bond), ops.ptr(conn), ops.ptr(mats), ops.ptr(dagg),
                                                  ops.ptr(dh), ops.ptr(dA), ops.ptr(ws), ws.numel(), B, N, E, D, Vb, ready,
                                                  _lib.stream_ptr()))


call(0)
for _ in range(5):
    call(1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters):
    call(1)
e1.record()
torch.cuda.synchronize()
print(f"batch {B} D {D}: {e0.elapsed_time(e1) / a.iters * 1e3:.1f} us per call (sorted order reused)")
