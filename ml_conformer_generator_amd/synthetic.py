"""Synthetic inputs for tests and benchmarks (no datasets or trained weights exist offline)."""
from __future__ import annotations

from typing import Sequence

import torch

DUMMY_CONTEXT = (53.6424, 108.3042, 151.4399)   # the reference's ONNX-export dummy (onnx_export_utils.py:16-18)


def synth_gcn_inputs(B: int, sizes: Sequence[int], seed: int):
    """Inputs built to the prepare_adj_mat_seer_input recipe (mol_utils.py:159-191):
    elements = zero-padded atomic numbers; dist_mat = distances + I (zero padded);
    adj_mat = {0,1} connectivity + I.  Geometry: a 1.45 A random walk."""
    g = torch.Generator().manual_seed(seed)
    Zs = torch.tensor([6, 7, 8, 9, 15, 16, 17, 35])
    el = torch.zeros(B, 42, dtype=torch.long)
    dm = torch.zeros(B, 42, 42)
    am = torch.zeros(B, 42, 42)
    for b, n in enumerate(sizes):
        steps = torch.randn(n, 3, generator=g)
        steps = 1.45 * steps / steps.norm(dim=1, keepdim=True)
        xyz = torch.cumsum(steps, 0).double()
        d = torch.sqrt(((xyz.unsqueeze(1) - xyz.unsqueeze(0)) ** 2).sum(-1))
        el[b, :n] = Zs[torch.randint(0, 3, (n,), generator=g)]
        dm[b, :n, :n] = d.float()
        dm[b] += torch.eye(42)
        conn = ((d < 1.8) & (d > 0)).float()
        am[b, :n, :n] = conn
        am[b] += torch.eye(42)
        am[b][am[b] > 0] = 1
    return el, dm, am
