#!/usr/bin/env python3
"""Latency at small batches (Streamlit-style calls: n_samples 4..40): ms per sampler step, host vs device."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ml_conformer_generator_amd import weights as W
from ml_conformer_generator_amd.egnn import EGNNDynamics
from ml_conformer_generator_amd.equivariant_diffusion import EquivariantDiffusion, PredefinedNoiseSchedule
dev = torch.device("cuda:0")
dyn = EGNNDynamics(device=dev); dyn.load_reference_state_dict(W.synth_edm_state_dict(1234))
dyn.set_precision(os.environ.get("MCG_DTYPE", "f32"))
T = 50
gm = EquivariantDiffusion(dynamics=dyn, in_node_nf=8, timesteps=1000, noise_precision=1e-5)
gm.gamma = PredefinedNoiseSchedule(timesteps=T, precision=1e-5); gm.T = T
for B, n in [tuple(int(v) for v in os.environ["MCG_SMALL_SHAPES"].split(",")[i:i+2]) for i in range(0, 2*len(os.environ["MCG_SMALL_SHAPES"].split(","))//2, 2)] if os.environ.get("MCG_SMALL_SHAPES") else ((4, 19), (16, 27), (40, 27), (64, 27)):
    nm = torch.ones(B, n, 1, device=dev); ctx = torch.zeros(B, n, 3, device=dev)
    gm(nm, None, ctx, 0); torch.cuda.synchronize()
    t0 = time.perf_counter(); x, h = gm(nm, None, ctx, 0); t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
    print(f"B={B} n={n}: host-issue {t_host/(T+1)*1e3:.3f} ms/step, wall {t_all/(T+1)*1e3:.3f} ms/step")
