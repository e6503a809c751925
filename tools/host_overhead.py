import sys, time; sys.path.insert(0, '.')
import torch, numpy as np
from ionic_mpnn_amd import model, synthetic, weights
dev = torch.device('cuda:0')
inp = synthetic.make_batch(4096, seed=0)
w = weights.init_weights("viscosity", 124, 72, num_steps=3, seed=1)
m = model.build_model(124, 72, num_steps=3, device=dev); m.load_weights(w)
d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
for _ in range(20): m.encode_pooled(d, fused=True)
torch.cuda.synchronize()
# host-only cost: tiny batch so the GPU is never the bottleneck
small = {k: v[:8].contiguous() for k, v in d.items()}
for _ in range(20): m.encode_pooled(small, fused=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500): m.encode_pooled(small, fused=True)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue per call: {(t1-t0)/500*1e6:.1f} us; incl. drain {(t2-t0)/500*1e6:.1f} us")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(300): m.encode_pooled(small, fused=True)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
