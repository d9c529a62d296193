"""Shape similarity of generated samples to the reference.

Mirrors the shape half of `evaluate_samples` (`cheminformatics/pipeline.py:30-85`): principal shape frames
(`get_shape_quadrupole_for_molecule`, `cheminformatics/shape_similarity.py:18-202`), the four-orientation search
and `tanimoto_score` (:448-492), batched over all candidates on the device.  The grid overlap is a HIP kernel
(`mcg_shape_tanimoto`); the frame construction is batched tensor algebra around it (neighbour cliques grown
level by level as index tensors, Gaussian inclusion-exclusion moments accumulated per molecule), with the 3x3
eigen-decomposition on the host in the reference's own LAPACK path so that eigenvector signs - which decide
between a frame and its mirror image - are the reference's.  The RDKit fingerprint similarity and mol-block
output of `evaluate_samples` are outside this build.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

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


# ----------------------------------------------------------------------------- principal shape frames
def _grow_cliques(adj: torch.Tensor, n_terms: int) -> Dict[int, torch.Tensor]:
    """All mutually-neighbouring index subsets i1 < ... < ik (k = 2..n_terms) of every molecule of the batch,
    as tensors [C_k, 1 + k] of (molecule, i1..ik) rows in lexicographic order (the reference enumerates them
    one molecule at a time with a Python backtracking search, shape_similarity.py:269-311)."""
    B, N, _ = adj.shape
    ar = torch.arange(N, device=adj.device)
    upper = adj & (ar.view(1, 1, N) > ar.view(1, N, 1))
    cur = torch.nonzero(upper)                                   # [C2, 3] = (b, i, j), i < j
    out = {2: cur}
    for k in range(3, n_terms + 1):
        if cur.shape[0] == 0:
            out[k] = cur.new_zeros((0, k + 1))
            continue
        nxt = []
        for lo in range(0, cur.shape[0], 1 << 18):              # bounded [chunk, N] candidate masks
            c = cur[lo: lo + (1 << 18)]
            cand = ar.view(1, N) > c[:, -1:].expand(-1, 1)
            for m in range(1, c.shape[1]):
                cand = cand & adj[c[:, 0], c[:, m]]
            rows, v = torch.nonzero(cand, as_tuple=True)
            nxt.append(torch.cat([c[rows], v.unsqueeze(1)], dim=1))
        cur = torch.cat(nxt, dim=0)
        out[k] = cur
    return out


def _shape_moments(points: torch.Tensor, real: torch.Tensor, cliques: Dict[int, torch.Tensor], alpha: float,
                   amplitude: float):
    """Gaussian-volume moments of every molecule with inclusion-exclusion signs (shape_similarity.py:36-129,
    integrals :337-402): volume [B], first moments [B,3], diagonal [B,3] and off-diagonal (xy, xz, yz) [B,3]
    second moments.  Terms are evaluated in fp32 like the reference; the per-molecule sums run in fp64."""
    B = points.shape[0]
    dev = points.device
    vol = torch.zeros(B, dtype=torch.float64, device=dev)
    first = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    ii = torch.zeros(B, 3, dtype=torch.float64, device=dev)
    ij = torch.zeros(B, 3, dtype=torch.float64, device=dev)

    def add(mol, c, a, amp, sign):
        k = (math.pi / a) ** 1.5
        w = (amp * k).to(torch.float32)
        off = torch.stack([c[:, 0] * c[:, 1], c[:, 0] * c[:, 2], c[:, 1] * c[:, 2]], dim=1)
        vol.index_add_(0, mol, sign * w.double())
        first.index_add_(0, mol, sign * (w.unsqueeze(1) * c).double())
        ii.index_add_(0, mol, sign * (w.unsqueeze(1) * (c ** 2 + 1 / (2 * a))).double())
        ij.index_add_(0, mol, sign * (w.unsqueeze(1) * off).double())

    mol1, at1 = torch.nonzero(real, as_tuple=True)
    add(mol1, points[mol1, at1], alpha, torch.full((mol1.shape[0],), amplitude, device=dev), 1.0)
    for order, idx in cliques.items():
        if idx.shape[0] == 0:
            continue
        centers = points[idx[:, :1], idx[:, 1:]]                                  # [C, order, 3]
        gamma = (centers ** 2).sum(-1).sum(-1) - (centers.sum(1) ** 2).sum(-1) / order     # :223-225
        amp = amplitude ** order * torch.exp(-alpha * gamma)
        add(idx[:, 0], centers.mean(1), order * alpha, amp, float((-1) ** (order - 1)))
    return vol, first, ii, ij


def _moment_tensor(ii: torch.Tensor, ij: torch.Tensor, vol: torch.Tensor) -> torch.Tensor:
    t = torch.stack([torch.stack([ii[:, 0], ij[:, 0], ij[:, 1]], 1), torch.stack([ij[:, 0], ii[:, 1], ij[:, 2]], 1),
                     torch.stack([ij[:, 1], ij[:, 2], ii[:, 2]], 1)], 1)
    return t.to(torch.float32) / vol.to(torch.float32).view(-1, 1, 1)


@torch.no_grad()
def shape_quadrupole_batch(coords: torch.Tensor, n_nodes: torch.Tensor, device=None, amplitude: float = AMPLITUDE,
                           atom_radius: float = ATOM_RADIUS, n_terms: int = 6,
                           neighbour_threshold: float = 2 * AMPLITUDE) -> Tuple[torch.Tensor, torch.Tensor]:
    """`get_shape_quadrupole_for_molecule` (shape_similarity.py:18-202) for a padded batch: principal moments
    [B,3] (descending) and coordinates in the principal shape frame [B,N,3] (zero on padded slots)."""
    dev = torch.device(device if device is not None else (coords.device if coords.is_cuda else "cuda:0"))
    x = coords.to(dev, torch.float32)
    B, N, _ = x.shape
    real = torch.arange(N, device=dev).unsqueeze(0) < n_nodes.to(dev).reshape(B, 1)
    alpha = get_alpha(atom_radius, amplitude)
    d = torch.sqrt(((x.unsqueeze(2) - x.unsqueeze(1)) ** 2).sum(-1))
    adj = (d < neighbour_threshold) & (d > 0) & real.unsqueeze(1) & real.unsqueeze(2)       # :244-260
    cliques = _grow_cliques(adj, n_terms)
    vol, first, _, _ = _shape_moments(x, real, cliques, alpha, amplitude)
    centred = (x - (first / vol.unsqueeze(1)).to(torch.float32).unsqueeze(1)) * real.unsqueeze(2)     # :86-89
    _, _, ii, ij = _shape_moments(centred, real, cliques, alpha, amplitude)
    _, vecs = torch.linalg.eigh(_moment_tensor(ii, ij, vol).cpu())                         # :143-144 (host LAPACK)
    rotated = (centred @ vecs.to(dev)) * real.unsqueeze(2)
    _, _, ii, ij = _shape_moments(rotated, real, cliques, alpha, amplitude)
    main = torch.diagonal(_moment_tensor(ii, ij, vol), dim1=1, dim2=2)                     # :184-199
    moments, order = torch.sort(main, dim=1, descending=True)
    frames = torch.gather(rotated, 2, order.unsqueeze(1).expand(B, N, 3))
    return moments, frames


@torch.no_grad()
def evaluate_shape(reference_coord: torch.Tensor, sample_coords: torch.Tensor, n_nodes: torch.Tensor, device=None
                   ) -> Tuple[torch.Tensor, List[dict]]:
    """Shape half of `evaluate_samples` (cheminformatics/pipeline.py:30-85) for heavy-atom coordinates:
    the reference in its principal frame, and per sample the best of the four orientations with its aligned
    coordinates.  `sample_coords` [B,N,3] padded, `n_nodes` [B]."""
    dev = torch.device(device if device is not None else (sample_coords.device if sample_coords.is_cuda else "cuda:0"))
    ref = reference_coord.to(torch.float32)
    ref = ref - ref.mean(0)                                                             # :38-39
    _, ref_pf = shape_quadrupole_batch(ref.unsqueeze(0), torch.tensor([ref.shape[0]]), dev)
    ref_pf = ref_pf[0].cpu()
    x = sample_coords.to(dev, torch.float32)
    B, N, _ = x.shape
    nn = n_nodes.to(dev).reshape(B)
    real = (torch.arange(N, device=dev).unsqueeze(0) < nn.unsqueeze(1)).unsqueeze(2)
    com = (x * real).sum(1, keepdim=True) / nn.view(B, 1, 1).clamp(min=1)             # :66-67
    _, frames = shape_quadrupole_batch((x - com) * real, nn, dev)
    scores, best, which = shape_tanimoto_batch(ref_pf, frames, nn, dev)
    pi = torch.pi
    rots = torch.stack([rotation_matrix(a) for a in (torch.zeros(3), torch.tensor([pi, 0, 0]),
                                                     torch.tensor([0, pi, 0]), torch.tensor([0, 0, pi]))]).to(dev)
    aligned = torch.einsum("bnk,bkj->bnj", frames, rots[which])
    aligned_c, best_c, which_c, nn_c = aligned.cpu(), best.cpu(), which.cpu(), nn.cpu()
    results = [{"coords": aligned_c[b, : int(nn_c[b])].clone(), "shape_tanimoto": float(best_c[b]),
                "orientation": int(which_c[b])} for b in range(B)]
    return ref_pf, results


def evaluate_samples(reference, samples, generator=None, device=None):
    """`evaluate_samples` of the reference (cheminformatics/pipeline.py:17-96) with its signature and return value:
    (mol block of the reference in its principal shape frame, per sample {"mol_block", "shape_tanimoto",
    "chemical_tanimoto"}).  The shape half - principal frames and the four-orientation Gaussian-volume Tanimoto search - runs on
    the HIP path for all samples at once (`evaluate_shape`); the chemical half (Morgan fingerprints radius 2, 2 048 bits, bond
    types on, chirality off, `TanimotoSimilarity`) and the mol-block output are RDKit's own and run ONLY where RDKit imports -
    this wrapper is untested offline (RDKit is absent from the build container and the GPU boxes;
    `tests/test_rdkit_optional.py` runs it where RDKit and a GPU exist)."""
    from rdkit import Chem
    from rdkit.Chem import rdFingerprintGenerator
    from rdkit.DataStructs.cDataStructs import TanimotoSimilarity
    from rdkit.Geometry import Point3D

    if generator is None:                                                       # pipeline.py:11-14
        generator = rdFingerprintGenerator.GetMorganGenerator(radius=2, fpSize=2048, includeChirality=False, useBondTypes=True)

    def set_positions(mol, coord):                                              # pipeline.py:99-105
        conf = mol.GetConformer()
        for i, point in enumerate(coord.tolist()):
            conf.SetAtomPosition(i, Point3D(point[0], point[1], point[2]))
        return mol

    reference = Chem.RemoveHs(reference)                                        # :32
    fp_ref = generator.GetFingerprint(reference)
    ref_coord = torch.tensor(reference.GetConformer().GetPositions(), dtype=torch.float32)
    stripped = [Chem.RemoveHs(s) for s in samples]                              # :61
    if not stripped:
        _, ref_pf = shape_quadrupole_batch((ref_coord - ref_coord.mean(0)).unsqueeze(0), torch.tensor([ref_coord.shape[0]]),
                                           torch.device(device if device is not None else "cuda:0"))
        return Chem.MolToMolBlock(set_positions(reference, ref_pf[0].cpu())), []
    n_nodes = torch.tensor([m.GetNumAtoms() for m in stripped])
    coords = torch.zeros(len(stripped), int(n_nodes.max()), 3)
    for b, m in enumerate(stripped):
        coords[b, : m.GetNumAtoms()] = torch.tensor(m.GetConformer().GetPositions(), dtype=torch.float32)
    ref_pf, shape = evaluate_shape(ref_coord, coords, n_nodes, device)
    ref_mol_block = Chem.MolToMolBlock(set_positions(reference, ref_pf))        # :44-45
    results = []
    for m, r in zip(stripped, shape):
        results.append({"mol_block": Chem.MolToMolBlock(set_positions(m, r["coords"])),
                        "shape_tanimoto": r["shape_tanimoto"],
                        "chemical_tanimoto": TanimotoSimilarity(fp_ref, generator.GetFingerprint(m))})
    return ref_mol_block, results
