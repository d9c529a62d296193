"""Oracle: tensor-only input construction (test infrastructure - see oracle/__init__.py).

Restates the RDKit-free helpers of utils/mol_utils.py on torch-CPU.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch


def masks_from_sizes(n_nodes: torch.Tensor, max_n_nodes: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """mol_utils.py:226-252 prepare_masks: prefix node mask [B,N,1]; edge mask
    [B*N*N,1] = outer product with the diagonal removed."""
    B = n_nodes.size(0)
    nm = torch.zeros(B, max_n_nodes)
    for b in range(B):
        nm[b, : int(n_nodes[b])] = 1
    em = nm.unsqueeze(1) * nm.unsqueeze(2)
    em = em * (~torch.eye(max_n_nodes, dtype=torch.bool)).unsqueeze(0)
    return nm.unsqueeze(2), em.reshape(B * max_n_nodes * max_n_nodes, 1)


def edm_input(n_samples: int, reference_context: torch.Tensor, context_norms: Dict[str, torch.Tensor],
              min_n_nodes: int, max_n_nodes: int):
    """mol_utils.py:255-295 prepare_edm_input: sizes from the CPU global RNG."""
    sizes = torch.randint(min_n_nodes, max_n_nodes + 1, (n_samples,))
    nm, em = masks_from_sizes(sizes, max_n_nodes)
    normed = (reference_context - context_norms["mean"]) / context_norms["mad"]
    ctx = normed.unsqueeze(0).repeat(n_samples, 1).unsqueeze(1).repeat(1, max_n_nodes, 1) * nm
    return nm, em, ctx


def inertia_tensor(coord: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """mol_utils.py:60-85.  NB the off-diagonal terms are NOT weighted (reference quirk)."""
    x, y, z = coord[:, 0], coord[:, 1], coord[:, 2]
    ixx = torch.sum(weights * (y ** 2 + z ** 2))
    iyy = torch.sum(weights * (x ** 2 + z ** 2))
    izz = torch.sum(weights * (x ** 2 + y ** 2))
    ixy, ixz, iyz = -torch.sum(x * y), -torch.sum(x * z), -torch.sum(y * z)
    return torch.tensor([[ixx, ixy, ixz], [ixy, iyy, iyz], [ixz, iyz, izz]], dtype=torch.float32)


def context_shape(coord: torch.Tensor):
    """mol_utils.py:88-107 get_context_shape: principal moments (unit masses)."""
    w = torch.ones(coord.size(0))
    _, vecs = torch.linalg.eigh(inertia_tensor(coord, w))
    rotated = torch.matmul(coord.to(torch.float32), vecs)
    return torch.diag(inertia_tensor(rotated, w)), rotated


def pairwise_distance(coord: torch.Tensor) -> torch.Tensor:
    """mol_utils.py:129-143 distance_matrix."""
    diff = coord.unsqueeze(1) - coord.unsqueeze(0)
    return torch.sqrt(torch.sum(torch.pow(diff, 2), 2))


def fragment_latent(coord: torch.Tensor, one_hot: torch.Tensor, n_samples: int,
                    max_n_nodes: int, min_n_nodes: int = 15):
    """Tensor half of mol_utils.py:298-342 prepare_fragment."""
    n = coord.size(0)
    if n >= min_n_nodes:
        raise ValueError("Fragment must contain fewer atoms than minimum generation size.")
    if n >= max_n_nodes:
        raise ValueError("Fragment has more atoms than the maximum number of atoms requested.")
    x = torch.nn.functional.pad(coord, (0, 0, 0, max_n_nodes - n))
    h = torch.nn.functional.pad(one_hot, (0, 0, 0, max_n_nodes - n))
    z_known = torch.cat([x.repeat(n_samples, 1, 1), h.repeat(n_samples, 1, 1)], dim=2)
    fixed = torch.zeros((n_samples, max_n_nodes, 1), dtype=torch.float32)
    fixed[:, :n, 0] = 1.0
    return z_known, fixed
