#!/usr/bin/env python3
"""Wall time of the stages of one generation pass at configs[1] (GPU box): sampler, hand-off, GCN, bond write-back,
D2H + molecule records."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ml_conformer_generator_amd import MLConformerGenerator, weights as W
from ml_conformer_generator_amd.synthetic import DUMMY_CONTEXT
from ml_conformer_generator_amd.handoff import prepare_adj_mat_seer_input_hip, bond_writeback_hip, molecules_from_tensors
dev = torch.device("cuda:0")
gen = MLConformerGenerator(diffusion_steps=100, device=dev, edm_weights=W.synth_edm_state_dict(1234, recipe="v2d"),
                           adj_mat_seer_weights=W.synth_adj_mat_seer_state_dict(4321))
ctx = torch.tensor(DUMMY_CONTEXT)
def sync(): torch.cuda.synchronize(dev)
for rep in range(3):
    t = [time.perf_counter()]
    x, h, node_mask = gen.edm_tensors(reference_context=ctx, n_samples=64, min_n_nodes=27, max_n_nodes=27, resample_steps=0,
                                      fixed_fragment=None, inertial_fragment_matching=True, blend_power=3, ifm_diffusion_level=50, sizes=None)
    sync(); t.append(time.perf_counter())
    n_nodes = node_mask.sum(1).reshape(-1).to(torch.long)
    el, dm, am = prepare_adj_mat_seer_input_hip(x, h, n_nodes, gen.dimension)
    sync(); t.append(time.perf_counter())
    bond = gen.adj_mat_seer.bond_orders(el, dm, am)
    sync(); t.append(time.perf_counter())
    sym, valid = bond_writeback_hip(bond, el, n_nodes)
    sync(); t.append(time.perf_counter())
    mols = molecules_from_tensors(x, el.to(torch.int8), sym, n_nodes.to(torch.int32), valid.to(torch.uint8))
    t.append(time.perf_counter())
    names = ["sampler (edm_tensors)", "hand-off", "GCN", "bond write-back", "D2H + molecule records"]
    print("  ".join(f"{n}: {1e3 * (b - a):.2f} ms" for n, a, b in zip(names, t, t[1:])), f" total {1e3 * (t[-1] - t[0]):.2f} ms")
