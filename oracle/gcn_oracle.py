"""Oracle: AdjMatSeer GCN adjacency pass (test infrastructure - see oracle/__init__.py).

Restates adj_mat_seer.py:32-57 (GraphConv) and :104-165 (AdjMatSeer.forward) over a
reference-layout state dict.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


def sym_norm(a: torch.Tensor) -> torch.Tensor:
    """adj_mat_seer.py:32-41: D^-1/2 A D^-1/2 with degree clamped at 1e-12."""
    inv = torch.rsqrt(a.sum(dim=-1).clamp(min=1e-12))
    return inv.unsqueeze(-1) * a * inv.unsqueeze(-2)


def graph_conv(sd: SD, name: str, x: torch.Tensor, L: torch.Tensor) -> torch.Tensor:
    """adj_mat_seer.py:43-57: propagate AFTER the affine map (bias is propagated too)."""
    return torch.bmm(L, F.linear(x, sd[name + ".linear.weight"], sd[name + ".linear.bias"]))


def adj_mat_seer(sd: SD, elements: torch.Tensor, dist_mat: torch.Tensor, adj_mat: torch.Tensor,
                 dimension: int = 42, embed: int = 64, n_bond: int = 5) -> torch.Tensor:
    """adj_mat_seer.py:104-165.  elements[B,42] i64, dist_mat/adj_mat[B,42,42] f32
    -> symmetric logits [B,42,42,5]."""
    B = elements.size(0)
    # distance-graph branch -> per-atom scalar bottleneck  (:115-125)
    L_dm = sym_norm(dist_mat)
    y = F.embedding(elements, sd["dm_nodes_embedding.weight"])
    for name in ("gcn1_dm", "gcn2_dm", "gcn3_dm"):
        y = F.relu(graph_conv(sd, name, y, L_dm))
    bottleneck = F.linear(y, sd["dm_resize.weight"], sd["dm_resize.bias"]).squeeze(-1)
    # main branch  (:130-152)
    scale = F.linear(bottleneck, sd["nodes_coord_fc.weight"], sd["nodes_coord_fc.bias"])
    v = F.embedding(elements, sd["nodes_embedding.weight"]) + scale.reshape(B, dimension, embed)
    L = sym_norm(adj_mat)
    for name in ("gcn1", "gcn2", "gcn3", "gcn4"):
        v = F.relu(graph_conv(sd, name, v, L))
    out = F.linear(v, sd["resize.weight"], sd["resize.bias"]).reshape(B, dimension, dimension, n_bond)
    return out.transpose(1, 2) + out                                            # :161-163


def bond_orders(logits: torch.Tensor) -> torch.Tensor:
    """The consumer's reduction (mol_utils.py:210-211): argmax over bond classes,
    strict lower triangle -> int64 [B,42,42]."""
    a = torch.argmax(logits, dim=-1)
    d = a.size(-1)
    return torch.tril(a, diagonal=-1) if a.dim() == 2 else a * torch.tril(torch.ones(d, d, dtype=a.dtype), diagonal=-1)
