"""Oracle: EGNN denoiser (test infrastructure - see oracle/__init__.py).

Functional restatement over a reference-layout state dict (keys as in
SURVEY.md section 8b).  Same op sequence as the reference: dense edge list,
gathered + concatenated [E, 842] edge inputs, per-edge Linear, scatter-add.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
NORM = 100.0  # egnn.py:15,92


def dense_edge_index(n_nodes: int, batch: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """egnn.py:515-541 - edge e = b*N*N + i*N + j, row = b*N+i, col = b*N+j."""
    node = torch.arange(n_nodes)
    r = node.repeat_interleave(n_nodes)
    c = node.repeat(n_nodes)
    off = (torch.arange(batch) * n_nodes).unsqueeze(1)
    return (r.unsqueeze(0) + off).reshape(-1).long(), (c.unsqueeze(0) + off).reshape(-1).long()


def pair_geometry(x: torch.Tensor, row: torch.Tensor, col: torch.Tensor):
    """egnn.py:404-415 coord2diff: squared distance and normalised difference."""
    d = x[row] - x[col]
    r2 = (d ** 2).sum(1, keepdim=True)
    return r2, d / torch.sqrt(r2 + 1e-8)


def segment_sum(data: torch.Tensor, row: torch.Tensor, n_seg: int) -> torch.Tensor:
    """egnn.py:418-437 unsorted_segment_sum with 'sum' normalisation (/100)."""
    acc = torch.zeros((n_seg, data.size(1)), dtype=data.dtype)
    acc.scatter_add_(0, row.unsqueeze(-1).expand_as(data), data)
    return acc / NORM


def masked_mean_removal(x: torch.Tensor, node_mask: torch.Tensor) -> torch.Tensor:
    """egnn.py:440-445 / equivariant_diffusion.py:48-53."""
    count = node_mask.sum(1, keepdim=True)
    return x - (x.sum(1, keepdim=True) / count) * node_mask


def gcl(sd: SD, p: str, h, row, col, edge_attr, node_mask, edge_mask):
    """egnn.py:38-85 GCL.forward (edge_model -> node_model -> mask)."""
    e_in = torch.cat([h[row], h[col], edge_attr], dim=1)                       # :45
    m = F.silu(F.linear(e_in, sd[p + "edge_mlp.0.weight"], sd[p + "edge_mlp.0.bias"]))
    m = F.silu(F.linear(m, sd[p + "edge_mlp.2.weight"], sd[p + "edge_mlp.2.bias"]))   # :46
    gate = torch.sigmoid(F.linear(m, sd[p + "att_mlp.0.weight"], sd[p + "att_mlp.0.bias"]))  # :48
    msg = m * gate * edge_mask                                                 # :49-51
    agg = segment_sum(msg, row, h.size(0))                                     # :59-64
    n_in = torch.cat([h, agg], dim=1)                                          # :66
    upd = F.linear(F.silu(F.linear(n_in, sd[p + "node_mlp.0.weight"], sd[p + "node_mlp.0.bias"])),
                   sd[p + "node_mlp.2.weight"], sd[p + "node_mlp.2.bias"])
    return (h + upd) * node_mask, m, msg, agg                                  # :67,83


def equivariant_update(sd: SD, p: str, h, x, row, col, unit_diff, edge_attr, node_mask, edge_mask):
    """egnn.py:111-149 EquivariantUpdate (no tanh / coords_range is unused)."""
    e_in = torch.cat([h[row], h[col], edge_attr], dim=1)                       # :122
    t = F.silu(F.linear(e_in, sd[p + "coord_mlp.0.weight"], sd[p + "coord_mlp.0.bias"]))
    t = F.silu(F.linear(t, sd[p + "coord_mlp.2.weight"], sd[p + "coord_mlp.2.bias"]))
    phi = F.linear(t, sd[p + "coord_mlp.4.weight"])                            # no bias :100
    trans = unit_diff * phi * edge_mask                                        # :124-127
    return (x + segment_sum(trans, row, x.size(0))) * node_mask                # :128-148


def equivariant_block(sd: SD, p: str, h, x, row, col, node_mask, edge_mask, d0):
    """egnn.py:188-222."""
    r2, unit = pair_geometry(x, row, col)                                      # :197
    ea = torch.cat([r2, d0], dim=1)                                            # :199
    h, *_ = gcl(sd, p + "gcl_0.", h, row, col, ea, node_mask, edge_mask)
    h, *_ = gcl(sd, p + "gcl_1.", h, row, col, ea, node_mask, edge_mask)
    x = equivariant_update(sd, p + "gcl_equiv.", h, x, row, col, unit, ea, node_mask, edge_mask)
    return h * node_mask, x                                                    # :221


def egnn(sd: SD, h, x, row, col, node_mask, edge_mask, n_blocks: int = 9, p: str = "dynamics.egnn."):
    """egnn.py:305-401."""
    d0, _ = pair_geometry(x, row, col)                                         # :313
    h = F.linear(h, sd[p + "embedding.weight"], sd[p + "embedding.bias"])      # :315
    for k in range(n_blocks):
        h, x = equivariant_block(sd, f"{p}e_block_{k}.", h, x, row, col, node_mask, edge_mask, d0)
    h = F.linear(h, sd[p + "embedding_out.weight"], sd[p + "embedding_out.bias"]) * node_mask
    return h, x


def egnn_dynamics(sd: SD, t, xh, node_mask, edge_mask, context, n_blocks: int = 9) -> torch.Tensor:
    """egnn.py:472-513 EGNNDynamics.forward: the 5-tensor operator seam.
    t[B,1], xh[B,N,11], node_mask[B,N,1], edge_mask[B*N*N,1], context[B,N,3] -> [B,N,11]."""
    B, N, _ = xh.shape
    row, col = dense_edge_index(N, B)
    nm = node_mask.reshape(B * N, 1)
    em = edge_mask.reshape(B * N * N, 1)
    flat = xh.reshape(B * N, -1) * nm                                          # :479
    x0 = flat[:, :3].clone()
    # time is broadcast to ALL nodes, padded ones included (:484-487)
    feats = torch.cat([flat[:, 3:], t.reshape(B, 1).repeat(1, N).reshape(B * N, 1),
                       context.reshape(B * N, -1)], dim=1)                     # :487-493
    h_out, x_out = egnn(sd, feats, x0, row, col, nm, em, n_blocks)
    vel = ((x_out - x0) * nm).reshape(B, N, 3)                                 # :499-507
    vel = masked_mean_removal(vel, node_mask.reshape(B, N, 1))                 # :509
    n_ctx = context.shape[-1]
    h_keep = h_out[:, : -(n_ctx + 1)].reshape(B, N, -1)                        # :503-505
    return torch.cat([vel, h_keep], dim=2)


def aggregate_standalone(m: torch.Tensor, gate: torch.Tensor, n_nodes: torch.Tensor) -> torch.Tensor:
    """Stand-alone form of the gate*mask*segment-sum (egnn.py:49-51,59-64,418-437) over
    the COMPACT real-edge list used by `mcg_egnn_aggregate`: molecule b contributes
    n_b*(n_b-1) rows ordered (i, j != i); output is [sum n_b, D] = sum_j m*gate / 100."""
    out = []
    off = 0
    for n in n_nodes.tolist():
        cnt = n * (n - 1)
        blk = (m[off:off + cnt] * gate[off:off + cnt].unsqueeze(1)).reshape(n, n - 1, -1)
        acc = torch.zeros(n, m.size(1))
        for j in range(n - 1):            # sequential j order == scatter_add_ order
            acc += blk[:, j]
        out.append(acc / NORM)
        off += cnt
    return torch.cat(out, 0)
