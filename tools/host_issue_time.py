import os, sys, time
sys.path.insert(0, ".")
import torch
from ml_conformer_generator_amd import weights as W, _lib
from ml_conformer_generator_amd.egnn import EGNNDynamics
dev = torch.device("cuda:0")
dyn = EGNNDynamics(device=dev); dyn.load_reference_state_dict(W.synth_edm_state_dict(1234))
B, n = 4, 19
plan = dyn.plan(torch.full((B,), n, dtype=torch.int32), n)
z = torch.randn(B, n, 11, device=dev); ctx = torch.zeros(B, n, 3, device=dev); t = torch.full((B,), 0.5, device=dev); out = torch.empty_like(z)
dyn.run(plan, t, z, ctx, out); torch.cuda.synchronize()
# host cost of ONE call when the queue is empty
ts = []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); dyn.run(plan, t, z, ctx, out); ts.append(time.perf_counter() - t0)
torch.cuda.synchronize()
print("graph" if os.environ.get("MCG_GRAPH", "1") != "0" else "plain", "host time per call (empty queue): %.3f ms" % (sorted(ts)[len(ts)//2] * 1e3))
