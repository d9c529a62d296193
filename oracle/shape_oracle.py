"""Oracle: Gaussian-volume shape Tanimoto on a grid (test infrastructure - see oracle/__init__.py).

Restates `tanimoto_score` and its helpers (cheminformatics/shape_similarity.py:327-334, 405-492) and the
principal shape frame that precedes it in `evaluate_samples` (`get_shape_quadrupole_for_molecule`,
shape_similarity.py:18-402: Gaussian inclusion-exclusion over mutually-neighbouring atom subsets up to order 6).
Pinned to tests/golden/shape_tanimoto.npz and shape_quadrupole.npz (generated from the reference).
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


# ------------------------------------------------------------------------- principal shape frame
def neighbour_matrix(coord: torch.Tensor, threshold: float) -> torch.Tensor:
    """shape_similarity.py:244-260: 0 < dist < threshold (no self loops)."""
    d = torch.sqrt(((coord.unsqueeze(1) - coord.unsqueeze(0)) ** 2).sum(2))
    return (d < threshold) & (d > 0)


def cliques_of_order(adj: torch.Tensor, order: int):
    """All index tuples i1 < i2 < ... < i_order whose members are mutual neighbours
    (find_r_cliques_fast, shape_similarity.py:269-311), in lexicographic order."""
    n = adj.size(0)
    out = []

    def grow(members, cand):
        if len(members) == order:
            out.append(list(members))
            return
        for v in cand:
            grow(members + [v], [u for u in cand if u > v and bool(adj[v, u])])

    grow([], list(range(n)))
    return torch.tensor(out, dtype=torch.long).reshape(-1, order)


def product_of_gaussians(centers: torch.Tensor, alpha: float, amplitude: float):
    """shape_similarity.py:205-230: centre, exponent and amplitude of a product of n equal Gaussians."""
    n = centers.size(1)
    gamma = (centers ** 2).sum(-1).sum(-1) - (centers.sum(1) ** 2).sum(-1) / n
    return centers.mean(1), n * alpha, amplitude ** n * torch.exp(-alpha * gamma)


def _moments(points: torch.Tensor, combos, alpha: float, amplitude: float):
    """Zeroth/first/second Gaussian-volume moments with inclusion-exclusion signs
    (shape_similarity.py:36-84 for volume/first, :88-129 for the second moments; integrals :337-402)."""
    def acc(c, a, amp):
        k = (math.pi / a) ** 1.5
        vol = (amp * k).sum() if torch.is_tensor(amp) else amp * k * c.size(0)
        amp_col = amp.unsqueeze(-1) if torch.is_tensor(amp) else amp
        first = (amp_col * c * k).sum(0)
        ii = (amp_col * k * (c ** 2 + 1 / (2 * a))).sum(0)
        ij = torch.stack([(amp_col.squeeze(-1) if torch.is_tensor(amp) else amp) * p * k for p in
                          (c[:, 0] * c[:, 1], c[:, 0] * c[:, 2], c[:, 1] * c[:, 2])]).sum(-1)
        return vol, first, ii, ij
    vol, first, ii, ij = acc(points, alpha, amplitude)
    for order, idx in combos.items():
        if idx.numel() == 0:
            continue
        c, a, amp = product_of_gaussians(points[idx], alpha, amplitude)
        v, f, d, o = acc(c, a, amp)
        sign = (-1) ** (order - 1)
        vol, first, ii, ij = vol + sign * v, first + sign * f, ii + sign * d, ij + sign * o
    return vol, first, ii, ij


def _tensor3(ii, ij, vol):
    return torch.tensor([[float(ii[0]), float(ij[0]), float(ij[1])],
                         [float(ij[0]), float(ii[1]), float(ij[2])],
                         [float(ij[1]), float(ij[2]), float(ii[2])]]) / vol


def shape_quadrupole(coordinates: torch.Tensor, amplitude: float = AMPLITUDE, atom_radius: float = ATOM_RADIUS,
                     n_terms: int = 6, neighbour_threshold: float = 2 * AMPLITUDE):
    """get_shape_quadrupole_for_molecule (shape_similarity.py:18-202): (principal moments, descending;
    coordinates in the principal shape frame)."""
    coordinates = coordinates.to(torch.float32)
    alpha = get_alpha(atom_radius, amplitude)
    adj = neighbour_matrix(coordinates, neighbour_threshold)
    combos = {k: cliques_of_order(adj, k) for k in range(2, n_terms + 1)}
    vol, first, _, _ = _moments(coordinates, combos, alpha, amplitude)
    centred = coordinates - first / vol                                  # :86-89
    _, _, ii, ij = _moments(centred, combos, alpha, amplitude)
    _, vecs = torch.linalg.eigh(_tensor3(ii, ij, vol))                  # :143-144
    rotated = centred @ vecs
    _, _, ii, ij = _moments(rotated, combos, alpha, amplitude)
    main = torch.diag(_tensor3(ii, ij, vol))                             # :184-199
    moments, order = torch.sort(main, descending=True)
    return moments, rotated[:, order]
