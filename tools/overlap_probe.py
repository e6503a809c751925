"""Do the plan kernels of one batch overlap the encoder kernel of another?  Times N x encoder alone, N x plan alone and
N x both (independent workspaces, two streams).  IMPNN_ENCODER_WORKGROUPS=224 leaves 32 CUs free for the plan kernels."""
import json, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from ionic_mpnn_amd import model as MM, synthetic, weights as W  # noqa: E402

B = 4096
dev = "cuda"
inp = synthetic.make_batch(B, seed=0)
m = MM.build_model(synthetic.DEFAULT_VA, synthetic.DEFAULT_VB, device=dev)
d = {k: torch.from_numpy(v).to(dev) for k, v in inp.items()}
plans = [m.plan_batch(d) for _ in range(2)]          # two slots planned
torch.cuda.synchronize()
pipe = m._pipeline
N = 200

def run_enc():
    for _ in range(N):
        m.encode_pooled(d, plan=plans[0])

def run_plan():
    for _ in range(N):
        pipe.next = 1                                 # always re-plan slot 1
        pipe.slots[1]["done"] = None
        m.plan_batch(d)

def timed(fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / N * 1e6

for _ in range(3):
    run_enc(); run_plan()
res = {"encoder_us": timed(run_enc), "plan_us": timed(run_plan)}
def both():
    for _ in range(N):
        pipe.next = 1
        pipe.slots[1]["done"] = None
        m.plan_batch(d)
        m.encode_pooled(d, plan=plans[0])
res["both_us"] = timed(both)
torch.cuda.synchronize()
t = time.perf_counter()
both()
res["both_host_enqueue_us"] = (time.perf_counter() - t) / N * 1e6   # host time to enqueue, GPU still running
torch.cuda.synchronize()
t = time.perf_counter()
run_enc()
res["enc_host_enqueue_us"] = (time.perf_counter() - t) / N * 1e6
torch.cuda.synchronize()
print(json.dumps(res))
