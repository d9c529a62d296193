"""Oracle: Gaussian-volume shape Tanimoto on a grid (test infrastructure - see oracle/__init__.py).

Restates `tanimoto_score` and its helpers (cheminformatics/shape_similarity.py:327-334, 405-492).
The principal-frame alignment that precedes it in `evaluate_samples` (clique enumeration,
shape_similarity.py:18-322) is host combinatorics and is not restated here.
"""
from __future__ import annotations

import math

import torch

ATOM_RADIUS = 1.60          # shape_similarity.py:14
AMPLITUDE = 2.70            # shape_similarity.py:15


def get_alpha(atom_radius: float = ATOM_RADIUS, amplitude: float = AMPLITUDE) -> float:
    """shape_similarity.py:327-334."""
    lam = 4 * math.pi / 3 / amplitude
    return (math.pi / lam ** (2 / 3)) / atom_radius ** 2


ALPHA = get_alpha()


def grid_axes(ref_coord: torch.Tensor, cand_coord: torch.Tensor, n: int = 40, bounds_scale: float = 6,
              max_sigma: float = ATOM_RADIUS):
    """The three linspace axes of the evaluation grid (shape_similarity.py:405-432, 476-480).
    Reference quirk reproduced: min/max are taken over dim=1 (the xyz axis) of the concatenated
    coordinates, so `min_coords[0..2]` are the per-atom minima of the FIRST THREE atoms, not a
    bounding box."""
    cat = torch.cat((ref_coord, cand_coord), dim=0)
    lo, _ = torch.min(cat, dim=1)
    hi, _ = torch.max(cat, dim=1)
    lo = lo - bounds_scale * max_sigma
    hi = hi + bounds_scale * max_sigma
    return [torch.linspace(lo[k], hi[k], n) for k in range(3)]


def density_on_grid(coord: torch.Tensor, points: torch.Tensor, alpha: float, amplitude: float) -> torch.Tensor:
    """shape_similarity.py:434-445: 1 - prod_atoms (1 - A exp(-alpha d^2))."""
    d2 = torch.cdist(points, coord) ** 2
    return 1 - torch.prod(1 - amplitude * torch.exp(-d2 * alpha), dim=-1)


def tanimoto_score(ref_coord: torch.Tensor, cand_coord: torch.Tensor, alpha: float = ALPHA,
                   amplitude: float = AMPLITUDE, n: int = 40) -> float:
    """shape_similarity.py:468-492."""
    xs, ys, zs = grid_axes(ref_coord, cand_coord, n)
    xg, yg, zg = torch.meshgrid(xs, ys, zs, indexing="ij")
    pts = torch.stack([xg.flatten(), yg.flatten(), zg.flatten()], dim=-1)
    f = density_on_grid(ref_coord, pts, alpha, amplitude)
    g = density_on_grid(cand_coord, pts, alpha, amplitude)
    fg = torch.sum(f * g)
    return float(fg / (torch.sum(f * f) + torch.sum(g * g) - fg))


def rotate_coord(coord: torch.Tensor, angles: torch.Tensor) -> torch.Tensor:
    """shape_similarity.py:448-463: coord @ Rx @ Ry @ Rz."""
    c, s = torch.cos(angles), torch.sin(angles)
    rx = torch.tensor([[1, 0, 0], [0, c[0], -s[0]], [0, s[0], c[0]]])
    ry = torch.tensor([[c[1], 0, s[1]], [0, 1, 0], [-s[1], 0, c[1]]])
    rz = torch.tensor([[c[2], -s[2], 0], [s[2], c[2], 0], [0, 0, 1]])
    return coord @ rx @ ry @ rz


def best_orientation_score(ref_coord: torch.Tensor, cand_coord: torch.Tensor):
    """The orientation search of evaluate_samples (cheminformatics/pipeline.py:48-85): identity plus
    the three pi-rotations; returns (best score, index 0..3)."""
    pi = torch.pi
    best, which = tanimoto_score(ref_coord, cand_coord), 0
    for k, ang in enumerate((torch.tensor([pi, 0, 0]), torch.tensor([0, pi, 0]), torch.tensor([0, 0, pi]))):
        sc = tanimoto_score(ref_coord, rotate_coord(cand_coord, ang))
        if sc > best:
            best, which = sc, k + 1
    return best, which
