"""Host-side input construction for the hot path (tensor-only, RDKit optional).

Mirrors the names/argument meaning of the reference's `utils/mol_utils.py` for
the functions that build the sampler's inputs; the RDKit-bound halves accept
either an RDKit Mol (when rdkit is importable) or plain (coordinates, atomic
numbers) so that the path runs on a box without RDKit.
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch

from .config import ATOMIC_NUMBERS, DIMENSION

_SYMBOL_TO_Z = {"H": 1, "C": 6, "N": 7, "O": 8, "F": 9, "P": 15, "S": 16, "Cl": 17, "Br": 35}
_Z_TO_CLASS = {z: i for i, z in enumerate(ATOMIC_NUMBERS)}


# ----------------------------------------------------------------------------- masks / context
def prepare_masks(n_nodes: torch.Tensor, max_n_nodes: int, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """node_mask [B,N,1] (prefix of ones) and edge_mask [B*N*N,1] (outer product
    minus diagonal).  Same contract as mol_utils.py:226-252, built without the
    per-sample Python loop."""
    n = n_nodes.reshape(-1).to(torch.long).cpu()
    node_mask = (torch.arange(max_n_nodes).unsqueeze(0) < n.unsqueeze(1)).to(torch.float32)
    edge_mask = node_mask.unsqueeze(1) * node_mask.unsqueeze(2)
    edge_mask = edge_mask * (1.0 - torch.eye(max_n_nodes)).unsqueeze(0)
    edge_mask = edge_mask.reshape(n.numel() * max_n_nodes * max_n_nodes, 1).to(device)
    node_mask = node_mask.unsqueeze(2).to(device)
    tag_canonical_masks(node_mask, edge_mask, n)
    return node_mask, edge_mask


def tag_canonical_masks(node_mask: torch.Tensor, edge_mask: torch.Tensor, sizes: torch.Tensor) -> None:
    """Masks built HERE are a prefix mask and its canonical edge mask by construction: say so on the tensor objects
    (sizes, a shared token, the version counters at tagging time) so that `EGNNDynamics.sizes_for` / `check_edge_mask` need
    no device compare + host sync for them - a later in-place edit bumps `_version` and voids the tag."""
    token = object()
    node_mask._mcg_mask_tag = (token, node_mask._version, sizes.to(torch.int32).reshape(-1).clone())
    edge_mask._mcg_mask_tag = (token, edge_mask._version)


def prepare_edm_input(n_samples: int, reference_context: torch.Tensor, context_norms: Dict[str, torch.Tensor],
                      min_n_nodes: int, max_n_nodes: int, device, sizes: torch.Tensor = None):
    """mol_utils.py:255-295.  Molecule sizes come from the CPU global RNG
    (`torch.randint`), exactly like the reference, so `torch.manual_seed` gives
    identical size draws.  `sizes` (optional, [n_samples] integers in [min, max]) hands in sizes drawn
    elsewhere - the sharded path draws the global vector once and gives every rank its slice."""
    if sizes is None:
        sizes = torch.randint(min_n_nodes, max_n_nodes + 1, (n_samples,))
    else:
        sizes = sizes.reshape(-1).to("cpu", torch.long)
        if sizes.numel() != n_samples or (n_samples and (int(sizes.min()) < min_n_nodes or int(sizes.max()) > max_n_nodes)):
            raise ValueError("sizes must hold n_samples values in [min_n_nodes, max_n_nodes]")
    node_mask, edge_mask = prepare_masks(sizes, max_n_nodes, device)
    normed = ((reference_context.cpu() - context_norms["mean"]) / context_norms["mad"]).to(device)
    ctx = normed.unsqueeze(0).repeat(n_samples, 1).unsqueeze(1).repeat(1, max_n_nodes, 1) * node_mask
    return node_mask, edge_mask, ctx


def get_moment_of_inertia_tensor(coord: torch.Tensor, weights: torch.Tensor) -> torch.Tensor:
    """mol_utils.py:60-85 (the off-diagonal products are unweighted there too)."""
    x, y, z = coord[:, 0], coord[:, 1], coord[:, 2]
    diag = [torch.sum(weights * (y * y + z * z)), torch.sum(weights * (x * x + z * z)),
            torch.sum(weights * (x * x + y * y))]
    xy, xz, yz = -torch.sum(x * y), -torch.sum(x * z), -torch.sum(y * z)
    return torch.tensor([[diag[0], xy, xz], [xy, diag[1], yz], [xz, yz, diag[2]]], dtype=torch.float32)


def get_context_shape(coord: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Principal moments of inertia with unit masses + coordinates in the principal
    frame (mol_utils.py:88-107)."""
    ones = torch.ones(coord.size(0))
    _, axes = torch.linalg.eigh(get_moment_of_inertia_tensor(coord, ones))
    rotated = torch.matmul(coord.to(torch.float32), axes)
    return torch.diag(get_moment_of_inertia_tensor(rotated, ones)), rotated


def distance_matrix(coordinates: torch.Tensor) -> torch.Tensor:
    """mol_utils.py:129-143."""
    delta = coordinates.unsqueeze(1) - coordinates.unsqueeze(0)
    return torch.sqrt(torch.sum(delta * delta, 2))


# ----------------------------------------------------------------------------- molecule I/O without RDKit
def parse_molblock_heavy_atoms(text: str) -> Tuple[torch.Tensor, List[int]]:
    """Heavy-atom coordinates [n,3] (float32) and atomic numbers of a V2000 MOL
    block - the subset of `Chem.RemoveHs(mol).GetConformer().GetPositions()`
    (conformer_generator.py:302-307) the path needs."""
    lines = text.splitlines()
    counts = lines[3]
    n_atoms = int(counts[0:3])
    xyz, zs = [], []
    for ln in lines[4:4 + n_atoms]:
        sym = ln[31:34].strip()
        if sym == "H":
            continue
        xyz.append([float(ln[0:10]), float(ln[10:20]), float(ln[20:30])])
        zs.append(_SYMBOL_TO_Z[sym])
    return torch.tensor(xyz, dtype=torch.float32), zs


def one_hot_classes(atomic_numbers: Sequence[int]) -> torch.Tensor:
    """Atom-class one-hot [n,8] int64 (molgraph.py:10,236-252)."""
    oh = torch.zeros(len(atomic_numbers), len(ATOMIC_NUMBERS), dtype=torch.long)
    for i, z in enumerate(atomic_numbers):
        oh[i, _Z_TO_CLASS[int(z)]] = 1
    return oh


def _fragment_xh(fixed_fragment, device):
    """(coords [n,3] f32, one-hot [n,8] i64) from an RDKit Mol or an (xyz, Z) pair
    (mol_utils.py:345-370 ifm_get_xh_from_fragment)."""
    if isinstance(fixed_fragment, (tuple, list)):
        xyz, zs = fixed_fragment
        xyz = torch.as_tensor(xyz, dtype=torch.float32)
        return xyz.to(device), one_hot_classes(list(zs)).to(device)
    from rdkit import Chem  # only reached when an RDKit Mol is handed in
    mol = Chem.RemoveAllHs(fixed_fragment)
    xyz = torch.tensor(mol.GetConformer().GetPositions(), dtype=torch.float32)
    zs = [a.GetAtomicNum() for a in mol.GetAtoms()]
    return xyz.to(device), one_hot_classes(zs).to(device)


ifm_get_xh_from_fragment = _fragment_xh


def prepare_fragment(n_samples: int, fixed_fragment, device, max_n_nodes: int = DIMENSION,
                     min_n_nodes: int = 15) -> Tuple[torch.Tensor, torch.Tensor]:
    """Latent z_known [B,N,11] and fixed_mask [B,N,1] for inpainting
    (mol_utils.py:298-342); same ValueErrors."""
    coord, h = _fragment_xh(fixed_fragment, device)
    n = coord.size(0)
    if n >= min_n_nodes:
        raise ValueError("Fragment must contain fewer atoms than minimum generation size.")
    if n >= max_n_nodes:
        raise ValueError("Fragment has more atoms than the maximum number of atoms requested.")
    z_known = torch.zeros(n_samples, max_n_nodes, 3 + h.size(1), dtype=torch.float32, device=device)
    z_known[:, :n, :3] = coord
    z_known[:, :n, 3:] = h.to(torch.float32)
    fixed_mask = torch.zeros((n_samples, max_n_nodes, 1), dtype=torch.float32, device=device)
    fixed_mask[:, :n, 0] = 1.0
    return z_known, fixed_mask


# ----------------------------------------------------------------------------- inertial fragment matching (f3)
def shift_moi_to_com_batch(moi_origin: torch.Tensor, r_coms: torch.Tensor, masses: torch.Tensor) -> torch.Tensor:
    """Inverse parallel-axis shift (mol_utils.py:527-550)."""
    B = r_coms.size(0)
    r = r_coms.view(B, 3, 1)
    r2 = (r_coms ** 2).sum(dim=1).view(B, 1, 1)
    eye = torch.eye(3, device=r_coms.device).expand(B, 3, 3)
    return moi_origin - masses.view(B, 1, 1) * (r2 * eye - r @ r.transpose(1, 2))


def ifm_prepare_gen_fragment_context(fixed_fragment_x, reference_context, context_norms, n_nodes,
                                     max_n_nodes: int, min_n_nodes: int, device):
    """Contexts for the separately generated fragments (mol_utils.py:373-457)."""
    B = n_nodes.size(0)
    n_ff = fixed_fragment_x.size(0)
    if n_ff >= min_n_nodes:
        raise ValueError("Fragment must contain fewer atoms than minimum generation size.")
    if n_ff >= max_n_nodes:
        raise ValueError("Fragment has more atoms than the maximum number of atoms requested.")
    ffx = fixed_fragment_x.cpu()
    moi_ff = get_moment_of_inertia_tensor(ffx, torch.ones(n_ff))
    moi_gen_origin = (torch.diag(reference_context.cpu()) - moi_ff).unsqueeze(0).repeat(B, 1, 1)
    gen_n = n_nodes.cpu().view(B, 1).float() - n_ff
    shift = (n_ff * ffx.mean(dim=0).view(1, 3)) / gen_n
    moi_gen_com = shift_moi_to_com_batch(moi_gen_origin, shift, gen_n)
    frag_ctx, rotation = torch.linalg.eigh(moi_gen_com)
    normed = ((frag_ctx - context_norms["mean"]) / context_norms["mad"]).to(device)
    n_frag_max = max_n_nodes - n_ff
    frag_node_mask, frag_edge_mask = prepare_masks(gen_n.long(), n_frag_max, device)
    ctx = normed.unsqueeze(1).repeat(1, n_frag_max, 1) * frag_node_mask
    return frag_node_mask, frag_edge_mask, ctx, shift.to(device), rotation.to(device)


def inverse_coord_transform(coord: torch.Tensor, shift: torch.Tensor, rotation: torch.Tensor) -> torch.Tensor:
    """Rotate back then translate (mol_utils.py:508-524)."""
    return torch.bmm(coord, rotation.transpose(1, 2)) - shift.view(coord.size(0), 1, 3)


def ifm_prepare_fragments_for_merge(fixed_fragment_x, fixed_fragment_h, gen_fragments_x, gen_fragments_h,
                                    device, max_n_nodes: int):
    """z_known = [fixed ; generated] per sample, fixed_mask on the first n_ff slots
    (mol_utils.py:460-505)."""
    B = gen_fragments_x.size(0)
    n_ff = fixed_fragment_x.size(0)
    x = torch.cat([fixed_fragment_x.unsqueeze(0).repeat(B, 1, 1).to(device), gen_fragments_x], dim=1)
    h = torch.cat([fixed_fragment_h.unsqueeze(0).repeat(B, 1, 1).to(device), gen_fragments_h], dim=1)
    fixed_mask = torch.zeros((B, max_n_nodes, 1), dtype=torch.float32, device=device)
    fixed_mask[:, :n_ff, 0] = 1.0
    return torch.cat([x, h], dim=2), fixed_mask


def ifm_merge_hip(fixed_fragment_x, fixed_fragment_h, gen_fragments_x, gen_fragments_h, shift, rotation, device,
                  max_n_nodes: int):
    """`inverse_coord_transform` + `ifm_prepare_fragments_for_merge` (mol_utils.py:508-524, :460-505) as ONE HIP
    launch (`mcg_ifm_merge`) on the first sampler run's device outputs: z_known [B,N,11], fixed_mask [B,N,1]."""
    from . import _lib
    f32 = dict(device=device, dtype=torch.float32)
    ffx = fixed_fragment_x.to(**f32).contiguous()
    ffh = fixed_fragment_h.to(**f32).contiguous()
    gx = gen_fragments_x.to(**f32).contiguous()
    gh = gen_fragments_h.to(**f32).contiguous()
    sh = shift.to(**f32).contiguous()
    rot = rotation.to(**f32).contiguous()
    B, n_gen, n_ff = int(gx.shape[0]), int(gx.shape[1]), int(ffx.shape[0])
    if n_ff + n_gen != max_n_nodes:
        raise ValueError("fixed fragment + generated fragment rows must fill max_n_nodes")
    z_known = torch.empty((B, max_n_nodes, 3 + ffh.shape[1]), **f32)
    fixed_mask = torch.empty((B, max_n_nodes, 1), **f32)
    _lib.check(_lib.lib().mcg_ifm_merge(_lib.dptr(ffx), _lib.dptr(ffh), n_ff, _lib.dptr(gx), _lib.dptr(gh), n_gen,
                                        _lib.dptr(sh), _lib.dptr(rot), B, max_n_nodes, _lib.dptr(z_known),
                                        _lib.dptr(fixed_mask), _lib.current_stream_ptr(torch.device(device))), "mcg_ifm_merge")
    return z_known, fixed_mask
