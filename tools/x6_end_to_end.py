#!/usr/bin/env python3
"""f32x6 against the exact-fp32 path END TO END (GPU box): the same seeded generation - size draw, T-step sampler,
hand-off, GCN, bond argmax - in both modes at configs[1] (64 x 27 atoms) and the configs[2] shape (256 ragged), and a
count of what differs: coordinates, atom types, adjacency (bond-order argmax of the real lower triangle).

    python tools/x6_end_to_end.py [--steps 100] [--recipe v2d|v2|legacy]

Synthetic weights: the network is untrained, so the sampler is not contractive - a 1e-6 difference per denoiser call
may or may not stay small over 100 calls; the script reports the measured growth, it does not assume it."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ml_conformer_generator_amd import MLConformerGenerator, weights as W
from ml_conformer_generator_amd.synthetic import DUMMY_CONTEXT


def run(gen, ctx, n_samples, n_atoms, variance, seed):
    torch.default_generator.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    gen._generate_shard(ctx, n_atoms, variance, None, n_samples, 0, None, True, 3, 50)
    b = gen.last_batch
    return {k: v.clone() for k, v in b.items()}


def compare(a, b):
    n = a["n_nodes"].cpu()
    B = n.numel()
    x0, x1 = a["x"].cpu(), b["x"].cpu()
    h0, h1 = a["h"].cpu().argmax(2), b["h"].cpu().argmax(2)
    bo0, bo1 = a["bond"].cpu(), b["bond"].cpu()
    N = x0.shape[1]
    real = torch.arange(N).unsqueeze(0) < n.unsqueeze(1)
    dx = ((x0 - x1).abs() * real.unsqueeze(2)).amax(dim=(1, 2))
    scale = float((x0.abs() * real.unsqueeze(2)).max())
    types_equal = ((h0 == h1) | ~real).all(1)
    D = bo0.shape[1]
    tri = torch.tril(torch.ones(D, D, dtype=torch.bool), -1).unsqueeze(0)
    inside = (torch.arange(D).view(1, D, 1) < n.view(B, 1, 1)) & (torch.arange(D).view(1, 1, D) < n.view(B, 1, 1))
    m = tri & inside
    adj_equal = ((bo0 == bo1) | ~m).flatten(1).all(1)
    return {"molecules": B, "max_abs_dx_over_max_abs_x": float(dx.max()) / scale,
            "median_abs_dx_over_max_abs_x": float(dx.median()) / scale,
            "molecules_with_equal_atom_types": int(types_equal.sum()),
            "molecules_with_equal_adjacency": int(adj_equal.sum()),
            "bond_entries_compared": int(m.sum()), "bond_entries_different": int(((bo0 != bo1) & m).sum()),
            "finite": bool(torch.isfinite(x0).all() and torch.isfinite(x1).all())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--recipe", default="v2d", help="v2d (bench weights) | v2 | legacy")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sd = (W.synth_edm_state_dict(1234, weight_gain=0.3) if a.recipe == "legacy" else W.synth_edm_state_dict(1234, recipe=a.recipe))
    gsd = W.synth_adj_mat_seer_state_dict(4321)
    ctx = torch.tensor(DUMMY_CONTEXT)
    gens = {m: MLConformerGenerator(diffusion_steps=a.steps, device=dev, edm_weights=sd, adj_mat_seer_weights=gsd,
                                    compute_dtype=m) for m in ("f32", "f32x6")}
    out = {"diffusion_steps": a.steps, "weights": a.recipe}
    for name, (n_samples, variance) in {"configs[1] 64 x 27": (64, 0), "configs[2] shape 256 ragged": (256, 12)}.items():
        res = {m: run(g, ctx, n_samples, 27, variance, 7) for m, g in gens.items()}
        again = run(gens["f32"], ctx, n_samples, 27, variance, 7)
        # control: the EXACT path with the context moved by one fp32 ulp - how far apart do two exact runs end?
        ctx_ulp = torch.nextafter(ctx, torch.full_like(ctx, float("inf")))
        nudged = run(gens["f32"], ctx_ulp, n_samples, 27, variance, 7)
        out[name] = {"f32x6_vs_f32": compare(res["f32"], res["f32x6"]), "f32_rerun_vs_f32": compare(res["f32"], again),
                     "f32_with_context_plus_one_ulp_vs_f32": compare(res["f32"], nudged)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
