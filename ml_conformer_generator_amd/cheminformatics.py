"""Shape similarity of generated samples to the reference (grid half).

Mirrors `tanimoto_score` / `rotate_coord` of the reference's `cheminformatics/shape_similarity.py`
(:448-492) and the orientation search of `evaluate_samples` (`cheminformatics/pipeline.py:64-85`), batched
over all candidates on the device.  Inputs are coordinates already expressed in their principal shape
frames; the reference's frame construction (`get_shape_quadrupole_for_molecule`, clique enumeration on the
host) and the RDKit fingerprint similarity are outside this build.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch

from . import _lib

ATOM_RADIUS = 1.60      # shape_similarity.py:14
AMPLITUDE = 2.70        # shape_similarity.py:15


def get_alpha(atom_radius: float = ATOM_RADIUS, gaussian_amplitude: float = AMPLITUDE) -> float:
    lam = 4 * math.pi / 3 / gaussian_amplitude
    return (math.pi / lam ** (2 / 3)) / atom_radius ** 2


ALPHA = get_alpha()


def rotation_matrix(angles: torch.Tensor) -> torch.Tensor:
    """M such that rotate_coord(coord, angles) == coord @ M  (shape_similarity.py:448-463)."""
    c, s = torch.cos(angles), torch.sin(angles)
    rx = torch.tensor([[1, 0, 0], [0, c[0], -s[0]], [0, s[0], c[0]]])
    ry = torch.tensor([[c[1], 0, s[1]], [0, 1, 0], [-s[1], 0, c[1]]])
    rz = torch.tensor([[c[2], -s[2], 0], [s[2], c[2], 0], [0, 0, 1]])
    return rx @ ry @ rz


def _grid_axes(ref_coord: torch.Tensor, n: int, bounds_scale: float = 6.0, max_sigma: float = ATOM_RADIUS):
    """The reference's `Grid` axes.  Its min/max run over the xyz axis of cat(ref, cand), so the bounds are
    the per-atom extrema of the first three atoms - which belong to the reference (shape_similarity.py:476-480,
    405-419); candidates never influence the grid."""
    if ref_coord.shape[0] < 3:
        raise ValueError("the reference needs at least 3 atoms")
    lo = ref_coord[:3].min(dim=1).values - bounds_scale * max_sigma
    hi = ref_coord[:3].max(dim=1).values + bounds_scale * max_sigma
    return torch.stack([torch.linspace(lo[k], hi[k], n) for k in range(3)])


@torch.no_grad()
def shape_tanimoto_batch(ref_coord: torch.Tensor, cand_coords: torch.Tensor, n_nodes: torch.Tensor,
                         device=None, n: int = 40, alpha: float = ALPHA, amplitude: float = AMPLITUDE
                         ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """scores[B,4] for the orientations (identity, pi about x, y, z), the best score per candidate and its
    orientation index - the search `evaluate_samples` runs one candidate at a time."""
    dev = torch.device(device if device is not None else (cand_coords.device if cand_coords.is_cuda else "cuda:0"))
    ref = ref_coord.to(torch.float32).cpu().contiguous()
    B, N, _ = cand_coords.shape
    pi = torch.pi
    rots = torch.stack([rotation_matrix(a) for a in (torch.zeros(3), torch.tensor([pi, 0, 0]),
                                                     torch.tensor([0, pi, 0]), torch.tensor([0, 0, pi]))])
    axes = _grid_axes(ref, n)
    ref_d, cand_d = ref.to(dev), cand_coords.to(dev, torch.float32).contiguous()
    nn = n_nodes.to(dev, torch.int32).contiguous()
    rots_d, axes_d = rots.to(dev, torch.float32).contiguous(), axes.to(dev).contiguous()
    f = torch.empty(n ** 3, device=dev)
    score = torch.empty(B * 4, device=dev)
    _lib.check(_lib.lib().mcg_shape_tanimoto(_lib.dptr(ref_d), ref.shape[0], _lib.dptr(cand_d), _lib.dptr(nn), B, N,
                                             _lib.dptr(rots_d), 4, _lib.dptr(axes_d), n, float(alpha), float(amplitude),
                                             _lib.dptr(f), _lib.dptr(score), _lib.current_stream_ptr(dev)),
               "mcg_shape_tanimoto")
    scores = score.reshape(B, 4)
    best, which = scores.max(dim=1)
    return scores, best, which


def tanimoto_score(ref_coord: torch.Tensor, cand_coord: torch.Tensor, alpha: float = ALPHA,
                   amplitude: float = AMPLITUDE, n: int = 40) -> float:
    """Single-pair form with the reference's signature (shape_similarity.py:468-492)."""
    cand = cand_coord.to(torch.float32).unsqueeze(0)
    scores, _, _ = shape_tanimoto_batch(ref_coord, cand, torch.tensor([cand.shape[1]]), n=n, alpha=alpha,
                                        amplitude=amplitude)
    return float(scores[0, 0])
