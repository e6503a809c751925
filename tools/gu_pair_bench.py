"""GatedUpdate alone on the chip (HIP events, one stream): the plain forward, the forward that keeps its activations, the
backward that recomputes and the backward that reads what was kept - with the executed-flop fraction of the f32 MFMA peak
(forward 12 D^2 per row; the backward's kernel 12 D^2 (kept) or 24 D^2 (recompute) + 12 D^2 in the weight-gradient GEMM).
python tools/gu_pair_bench.py [--dim 128] [--rows 85000]"""
import argparse, json, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ionic_mpnn_amd import ops, autograd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--rows", type=int, default=85000)
a = ap.parse_args()
D, rows = a.dim, a.rows
PEAK = 157.3e12
g = torch.Generator().manual_seed(0)
mk = lambda *s: (torch.randn(*s, generator=g) / (s[0] ** 0.5 if len(s) > 1 else 1)).cuda()
W = [mk(2 * D, D), mk(D), mk(2 * D, D), mk(D), mk(2 * D, D), mk(D), torch.ones(D).cuda(), torch.zeros(D).cuda()]
h, agg, go = (torch.randn(rows, D, generator=g).cuda() for _ in range(3))
ts = (h, agg, *W)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


res = {"dim": D, "rows": rows}
res["fwd_us"] = timed(lambda: ops.gated_update(*ts, 1e-3))
res["fwd_keep_us"] = timed(lambda: ops.gated_update(*ts, 1e-3, save=True))
res["bwd_recompute_us"] = timed(lambda: autograd._gated_update_backward(ts, 1e-3, go))


def pair():
    _, kept = ops.gated_update(*ts, 1e-3, save=True)
    autograd._gated_update_backward(ts, 1e-3, go, None, kept)


res["bwd_kept_us"] = timed(pair) - res["fwd_keep_us"]
fl = 12 * D * D * rows
res["fwd_frac_of_f32_peak"] = fl / (res["fwd_keep_us"] * 1e-6) / PEAK
res["bwd_kept_frac_of_f32_peak"] = 2 * fl / (res["bwd_kept_us"] * 1e-6) / PEAK
res["note"] = "backward times include the weight-gradient GEMM and the partial-sum reduction behind the main kernel"
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in res.items()}))
