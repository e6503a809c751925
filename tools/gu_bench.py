"""GatedUpdate forward / backward timings at a given width and row count (layer entries, HIP events)."""
import argparse, json, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ionic_mpnn_amd import ops, autograd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dim", type=int, default=128)
ap.add_argument("--rows", type=int, nargs="+", default=[1280, 163840])
args = ap.parse_args()
D = args.dim
g = torch.Generator().manual_seed(0)
mk = lambda *s: (torch.randn(*s, generator=g) / (s[0] ** 0.5 if len(s) > 1 else 1)).cuda()
W = [mk(2 * D, D), mk(D), mk(2 * D, D), mk(D), mk(2 * D, D), mk(D), torch.ones(D).cuda(), torch.zeros(D).cuda()]
for rows in args.rows:
    h, agg = torch.randn(rows // 40, 40, D, generator=g).cuda(), torch.randn(rows // 40, 40, D, generator=g).cuda()
    def fwd():
        return ops.gated_update(h, agg, *W, 1e-3)
    hg = h.clone().requires_grad_(True)
    Wg = [w.clone().requires_grad_(True) for w in W]
    out = autograd.GatedUpdate.apply(hg, agg, *Wg, 1e-3)
    go = torch.randn_like(out)
    def bwd():
        torch.autograd.grad(out, [hg] + Wg, go, retain_graph=True)
    res = {"dim": D, "rows": rows}
    for name, fn in (("fwd_us", fwd), ("bwd_us", bwd)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 30 * 1e3
    print(json.dumps(res))
