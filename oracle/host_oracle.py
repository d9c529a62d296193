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


# ------------------------------------------------------------------------------------------------
# EDM -> GCN hand-off, bond write-back and the tensor work between the two sampler runs of inertial
# fragment matching.  The RDKit-owned decisions inside these
# reference functions cannot run here; each substitute is named where it is made.
# ------------------------------------------------------------------------------------------------
_ATOM_DECODER = {0: "C", 1: "N", 2: "O", 3: "F", 4: "P", 5: "S", 6: "Cl", 7: "Br"}          # utils/config.py:9-18
_ATOMIC_NUMBER = {"C": 6, "N": 7, "O": 8, "F": 9, "P": 15, "S": 16, "Cl": 17, "Br": 35}
# single-bond covalent radii (Angstrom, Cordero 2008) - the radii table of the connectivity SUBSTITUTE below
_RCOV = {6: 0.76, 7: 0.71, 8: 0.66, 9: 0.57, 15: 1.07, 16: 1.05, 17: 1.02, 35: 1.20}
_MAX_VALENCE = {6: 4, 7: 4, 8: 2, 9: 1, 15: 5, 16: 6, 17: 1, 35: 1}
_BOND_VALENCE2 = (0, 2, 4, 6, 3)          # twice the valence of bond classes none / 1 / 2 / 3 / aromatic


def adj_mat_seer_input(positions: torch.Tensor, one_hot: torch.Tensor, n_nodes: torch.Tensor, dimension: int = 42,
                       cov_factor: float = 1.3, order=None, conn=None, return_coords: bool = False):
    """Tensor half of `samples_to_rdkit_mol` (mol_utils.py:18-57) + `prepare_adj_mat_seer_input` (:146-194),
    one molecule at a time like the reference:
      * atoms = argmax(one_hot) -> element symbol (:41-45); the coordinates travel through the XYZ text block
        written with "%.9f" (:46-51) and come back as DOUBLES (`torch.tensor(conf.GetPositions())`, :171);
      * [`canonicalise`, :110-126] `RenumberAtoms(mol, order)`: the atom at position p of everything below is
        generation atom order[b][p];
      * elements = atomic numbers, zero padded (`MolGraph.elements_vector`, molgraph.py:224-234);
      * dist_mat = fp64 `distance_matrix` (:129-143), zero padded to `dimension`, + I, stored into the fp32
        batch tensor (:178-187);
      * adj_mat = (1-order connectivity > 0) + I clamped to {0, 1} (:175-180).
    The two RDKit-owned decisions are ARGUMENTS: `order` (list of per-molecule permutations, the value of
    `_smilesAtomOutputOrder`, :118-124) and `conn` (list of per-molecule {0,1} [n,n] connectivities in GENERATION order,
    what `DetermineConnectivity` perceives before the renumbering, :117).  Left None they take the SUBSTITUTES (parity
    unpinned, RDKit absent): the covalent-radius rule d_ij < cov_factor * (r_i + r_j) that RDKit's
    `DetermineConnectivity` documents, and generation order.
    `return_coords`: also the fp64 coordinates in the order of the GCN input (the `canonicalised_samples`)."""
    B = positions.size(0)
    elements = torch.zeros(B, dimension, dtype=torch.long)
    dist_b = torch.zeros(B, dimension, dimension)
    adj_b = torch.zeros(B, dimension, dimension)
    coords_out = []
    for b in range(B):
        n = int(n_nodes[b])
        perm = list(range(n)) if order is None or order[b] is None else [int(v) for v in order[b]][:n]
        assert sorted(perm) == list(range(n)), "order must be a permutation of the molecule's atoms"
        atoms = torch.argmax(one_hot[b], dim=1)
        z_gen = [_ATOMIC_NUMBER[_ATOM_DECODER[int(atoms[i])]] for i in range(n)]
        coord_gen = torch.tensor([[float("%.9f" % float(positions[b, i, k])) for k in range(3)] for i in range(n)],
                                 dtype=torch.float64).reshape(n, 3)
        if conn is None or conn[b] is None:
            r = torch.tensor([_RCOV[v] for v in z_gen], dtype=torch.float64)
            conn_gen = (pairwise_distance(coord_gen) < cov_factor * (r.unsqueeze(0) + r.unsqueeze(1)))
        else:
            conn_gen = torch.as_tensor(conn[b])[:n, :n] != 0
        conn_gen = conn_gen & ~torch.eye(n, dtype=torch.bool)
        idx = torch.tensor(perm, dtype=torch.long)
        z = [z_gen[i] for i in perm]                                   # RenumberAtoms: new atom p = old atom order[p]
        coord = coord_gen[idx].reshape(n, 3)
        dist = pairwise_distance(coord)
        pad = torch.nn.functional.pad(dist, (0, dimension - n, 0, dimension - n), "constant", 0) + torch.eye(dimension)
        sc = torch.zeros(dimension, dimension)
        sc[:n, :n] = conn_gen[idx][:, idx].float()
        sc = sc + torch.eye(dimension)
        sc[sc > 0] = 1
        elements[b, :n] = torch.tensor(z, dtype=torch.long)
        dist_b[b] = pad                     # fp64 -> fp32 on assignment, as in the reference
        adj_b[b] = sc
        coords_out.append(coord)
    if return_coords:
        return elements, dist_b, adj_b, coords_out
    return elements, dist_b, adj_b


def bond_writeback(bond_argmax: torch.Tensor, elements: torch.Tensor, n_nodes: torch.Tensor):
    """`redefine_bonds` (mol_utils.py:197-223), tensor half: `tril(argmax)` with the diagonal removed (:210-211),
    one bond (i, j) per non-zero entry with i, j < n (:213-220) - returned mirrored.  Then the validity
    SUBSTITUTE for `standardize_mol(...) is not None` (conformer_generator.py:362-366; standardizer.py:83-111 is
    RDKit sanitisation + MMFF): no atom above its maximum valence, one connected fragment.  Plain loops."""
    B, D, _ = bond_argmax.shape
    sym = torch.zeros(B, D, D, dtype=torch.int8)
    valid = torch.zeros(B, dtype=torch.bool)
    for b in range(B):
        n = int(n_nodes[b])
        repr_m = torch.tril(bond_argmax[b].to(torch.long))
        repr_m = repr_m * (1 - torch.eye(D, dtype=torch.long))
        nbrs = [[] for _ in range(n)]
        val2 = [0] * n
        for i in range(n):
            for j in range(n):
                t = int(repr_m[i, j])
                if t != 0:
                    sym[b, i, j] = t
                    sym[b, j, i] = t
                    nbrs[i].append(j); nbrs[j].append(i)
                    val2[i] += _BOND_VALENCE2[t]; val2[j] += _BOND_VALENCE2[t]
        if n == 0:
            continue
        ok = all(val2[i] <= 2 * _MAX_VALENCE.get(int(elements[b, i]), 0) for i in range(n))
        seen, stack = {0}, [0]
        while stack:
            for j in nbrs[stack.pop()]:
                if j not in seen:
                    seen.add(j); stack.append(j)
        valid[b] = ok and len(seen) == n
    return sym, valid


def ifm_merge_input(fixed_fragment_x: torch.Tensor, fixed_fragment_h: torch.Tensor, gen_fragments_x: torch.Tensor,
                    gen_fragments_h: torch.Tensor, shift: torch.Tensor, rotation: torch.Tensor, max_n_nodes: int):
    """`inverse_coord_transform` (mol_utils.py:508-524: bmm with the transposed rotation, then minus the shift)
    followed by `ifm_prepare_fragments_for_merge` (:460-505: fixed fragment first, cat along atoms, cat along
    channels, fixed_mask on the first n_ff rows)."""
    B = gen_fragments_x.size(0)
    x_rot = torch.bmm(gen_fragments_x, torch.transpose(rotation, 1, 2)) - shift.view(B, 1, 3)
    n_ff = fixed_fragment_x.size(0)
    x_prep = torch.cat([fixed_fragment_x.unsqueeze(0).repeat(B, 1, 1), x_rot], dim=1)
    h_prep = torch.cat([fixed_fragment_h.unsqueeze(0).repeat(B, 1, 1), gen_fragments_h], dim=1)
    z_known = torch.cat([x_prep, h_prep], dim=2)
    fixed_mask = torch.zeros((B, max_n_nodes, 1), dtype=torch.float32)
    fixed_mask[:, :n_ff, 0] = 1.0
    return z_known, fixed_mask
