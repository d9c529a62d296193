import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from ml_conformer_generator_amd import MLConformerGenerator, weights as W
from ml_conformer_generator_amd.synthetic import DUMMY_CONTEXT
dev = torch.device("cuda:0")
gsd = W.synth_adj_mat_seer_state_dict(4321)
ctx = torch.tensor(DUMMY_CONTEXT)
for name, sd in (("v2", W.synth_edm_state_dict(1234)), ("v2d", W.synth_edm_state_dict(1234, recipe="v2d")), ("legacy0.3", W.synth_edm_state_dict(1234, weight_gain=0.3))):
    for T in (20, 100):
        gen = MLConformerGenerator(diffusion_steps=T, device=dev, edm_weights=sd, adj_mat_seer_weights=gsd)
        gm = gen.generative_model
        gm.trace = []
        torch.default_generator.manual_seed(7); torch.cuda.manual_seed(7)
        gen._generate_shard(ctx, 27, 0, None, 64, 0, None, True, 3, 50)
        tr = gm.trace
        mx = [float(z.abs().max()) for z in tr]
        fin = [bool(torch.isfinite(z).all()) for z in tr]
        first_bad = next((i for i, f in enumerate(fin) if not f), None)
        x = gen.last_batch["x"]
        print(name, "T", T, "steps", len(tr), "first non-finite step", first_bad, "max|z| at 0/1/mid/last", mx[0], mx[1] if len(mx) > 1 else None, mx[len(mx)//2], mx[-1], "x finite", bool(torch.isfinite(x).all()), "max|x|", float(x.abs().max()))
        gm.trace = None
