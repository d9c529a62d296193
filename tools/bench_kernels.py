#!/usr/bin/env python3
"""Micro-benchmarks on the GPU box: edge kernel per launch, one full denoiser call (phi).
Usage: python tools/bench_kernels.py [--shape c2|c3] [--mt 0|1|4] [--four-tile-units n|-1|all] [--ranges n] [--latency-mode m]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ml_conformer_generator_amd import _lib, weights as W
from ml_conformer_generator_amd.egnn import EGNNDynamics

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="c2")
ap.add_argument("--mt", type=int, default=0)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--mols", type=int, default=0, help="override: this many molecules of --atoms atoms")
ap.add_argument("--atoms", type=int, default=27)
ap.add_argument("--four-tile-units", default="0", help="mcg_plan_opts.four_tile_units: 0 auto, -1 none, n, or 'all'")
ap.add_argument("--ranges", type=int, default=0, help="mcg_plan_opts.n_ranges (0 = the library's choice)")
ap.add_argument("--latency-mode", type=int, default=-1, help="mcg_plan_set_latency_mode: -1 auto, 0 four-tile units only, 1 k_edge_ns")
ap.add_argument("--gemm-rn", type=int, default=0, help="mcg_egnn_set_option(MCG_OPT_GEMM_RN): wave tile width of the 32-row GEMM kernels")
ap.add_argument("--bf16-lds", type=int, default=0, help="mcg_egnn_set_option(MCG_OPT_GEMM_BF16_LDS): 0 auto, 1 the 32-row kernel, 2 the LDS-staged kernel")
ap.add_argument("--node-fused", type=int, default=0, help="mcg_egnn_set_option(MCG_OPT_NODE_FUSED): 0 auto, 1 three launches per layer, 2 one fused launch")
a = ap.parse_args()
dev = torch.device("cuda:0")
dyn = EGNNDynamics(device=dev)
dyn.load_reference_state_dict(W.synth_edm_state_dict(1234))
dyn.set_precision(a.dtype)
if a.gemm_rn:
    dyn.set_option(_lib.OPT_GEMM_RN, a.gemm_rn)
if a.bf16_lds:
    dyn.set_option(_lib.OPT_GEMM_BF16_LDS, a.bf16_lds)
if a.node_fused:
    dyn.set_option(_lib.OPT_NODE_FUSED, a.node_fused)
if a.mols > 0:
    sizes = torch.full((a.mols,), a.atoms, dtype=torch.int32); N = a.atoms
elif a.shape == "c2":
    sizes = torch.full((64,), 27, dtype=torch.int32); N = 27
else:
    torch.manual_seed(7); sizes = torch.randint(15, 40, (256,)).to(torch.int32); N = 39
ftu = _lib.ALL_FOUR_TILE if a.four_tile_units == "all" else int(a.four_tile_units)
plan = dyn.plan(sizes, N, edge_mt=a.mt, four_tile_units=ftu, n_ranges=a.ranges)
if a.latency_mode >= 0:
    plan.set_latency_mode(a.latency_mode)
B = sizes.numel()
z = torch.randn(B, N, 11, device=dev); ctx = torch.zeros(B, N, 3, device=dev); t = torch.full((B,), 0.5, device=dev)
out = dyn.run(plan, t, z, ctx)
L = _lib.lib(); st = _lib.current_stream_ptr(dev)
def timed(fn, iters):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn_iters = fn(iters) if False else [fn() for _ in range(iters)]; e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
ms_edge = timed(lambda: L.mcg_bench_edge(dyn.handle, plan.handle, 4, 0, 10, st), 5) / 10
ms_eq = timed(lambda: L.mcg_bench_edge(dyn.handle, plan.handle, 4, 1, 10, st), 5) / 10
ms_phi = timed(lambda: dyn.run(plan, t, z, ctx, out), a.iters)
E = plan.n_real_edges
fl = 2.0 * E * (420 * 420 + 3 * 420)
print(f"dtype={a.dtype} shape={a.shape} mt={plan.edge_mt} waves={plan.n_edge_waves} E={E} M={plan.n_real_nodes} "
      f"edge_gcl={ms_edge*1e3:.1f}us ({fl/ms_edge/1e9:.1f} TF/s) edge_equiv={ms_eq*1e3:.1f}us phi={ms_phi:.3f}ms")
