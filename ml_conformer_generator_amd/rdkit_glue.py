"""RDKit-owned stages around the hot path - used only when `import rdkit` succeeds.

Mirrors the reference's host glue so that a user WITH RDKit gets the reference's own
validity gate and return type (`List[Chem.Mol]`):
    samples_to_rdkit_mol / canonicalise / prepare_adj_mat_seer_input / redefine_bonds
        (utils/mol_utils.py:18-57, 110-126, 146-194, 197-223)
    standardize_mol (utils/standardizer.py:62-111)
RDKit is NOT installed in the build container or on the GPU box, so this module is not
exercised by the test suite: parity unpinned at the RDKit boundary.
Differences from the reference are confined to data movement: samples are copied to the host
once (the reference syncs the device once per printed coordinate) and the bond argmax comes
back from the device as int8.
"""
from __future__ import annotations

from typing import List

import torch
from rdkit import Chem  # noqa: F401 - import error here means "use the native hand-off"
from rdkit.Chem import AllChem, rdDetermineBonds
from rdkit.Chem.MolStandardize import rdMolStandardize

from .config import DIMENSION

_BOND_TYPES = {1: Chem.rdchem.BondType.SINGLE, 2: Chem.rdchem.BondType.DOUBLE,
               3: Chem.rdchem.BondType.TRIPLE, 4: Chem.rdchem.BondType.AROMATIC}


def samples_to_rdkit_mol(positions, one_hot, node_mask=None, atom_decoder=None) -> List["Chem.Mol"]:
    """XYZ block per sample -> Chem.MolFromXYZBlock (mol_utils.py:18-57)."""
    pos = positions.detach().cpu()
    types = torch.argmax(one_hot.detach().cpu(), dim=2)
    counts = (node_mask.detach().cpu().sum(dim=1).reshape(-1).to(torch.long) if node_mask is not None
              else torch.full((pos.shape[0],), pos.shape[1], dtype=torch.long))
    mols = []
    for b in range(pos.shape[0]):
        n = int(counts[b])
        lines = ["%d\n\n" % n]
        for i in range(n):
            lines.append("%s %.9f %.9f %.9f\n" % (atom_decoder[int(types[b, i])], pos[b, i, 0], pos[b, i, 1], pos[b, i, 2]))
        mol = Chem.MolFromXYZBlock("".join(lines))
        if mol is not None:
            mols.append(mol)
    return mols


def canonicalise(mol):
    """1-order connectivity guess + SMILES output order (mol_utils.py:110-126)."""
    rdDetermineBonds.DetermineConnectivity(mol)
    Chem.MolToSmiles(mol)
    order = [int(tok) for tok in mol.GetProp("_smilesAtomOutputOrder").strip("[]").split(",") if tok != ""]
    return Chem.RenumberAtoms(mol, order)


def prepare_adj_mat_seer_input(mols, dimension: int, device):
    """elements / dist_mat / adj_mat batches + canonicalised mols (mol_utils.py:146-194)."""
    n_s = len(mols)
    elements = torch.zeros(n_s, dimension, dtype=torch.long)
    dist = torch.zeros(n_s, dimension, dimension)
    adj = torch.zeros(n_s, dimension, dimension)
    ordered = []
    eye = torch.eye(dimension)
    for k, raw in enumerate(mols):
        mol = canonicalise(raw)
        xyz = torch.tensor(mol.GetConformer().GetPositions())
        n = mol.GetNumAtoms()
        for atom in mol.GetAtoms():
            elements[k, atom.GetIdx()] = atom.GetAtomicNum()
        conn = torch.zeros(dimension, dimension)
        for bond in mol.GetBonds():
            i, j = bond.GetBeginAtomIdx(), bond.GetEndAtomIdx()
            conn[i, j] = conn[j, i] = 1.0
        adj[k] = torch.clamp(conn + eye, max=1.0)
        d = torch.sqrt(((xyz.unsqueeze(1) - xyz.unsqueeze(0)) ** 2).sum(-1))
        dist[k, :n, :n] = d.to(torch.float32)
        dist[k] += eye
        ordered.append(mol)
    return elements.to(device), dist.to(device), adj.to(device), ordered


def redefine_bonds(mol, bond_orders: torch.Tensor):
    """Strip all bonds, then add those of the strict lower triangle of the argmax (mol_utils.py:197-223)."""
    n = mol.GetNumAtoms()
    bare = Chem.MolFromXYZBlock(Chem.MolToXYZBlock(mol))
    editable = Chem.EditableMol(bare)
    low = torch.tril(bond_orders.to(torch.long), diagonal=-1)
    for i in range(n):
        for j in range(i):
            order = int(low[i, j])
            if order != 0:
                editable.AddBond(i, j, _BOND_TYPES[order])
    return editable.GetMol()


def _tartrate_flattened(m):
    patt = Chem.MolFromSmarts("OC(=O)C(O)C(O)C(=O)O")
    params = Chem.AdjustQueryParameters.NoAdjustments()
    params.adjustDegree = True
    params.adjustDegreeFlags = Chem.AdjustQueryWhichFlags.ADJUST_IGNORENONE
    hits = m.GetSubstructMatches(Chem.AdjustQueryProperties(patt, params))
    if hits:
        m = Chem.Mol(m)
        for hit in hits:
            for idx in (hit[3], hit[5]):
                m.GetAtomWithIdx(idx).SetChiralTag(Chem.ChiralType.CHI_UNSPECIFIED)
    return m


def standardize_mol(mol, optimize_geometry: bool = True):
    """The validity gate (standardizer.py:83-111): any failure -> None."""
    try:
        m = rdMolStandardize.FragmentParent(mol)
        Chem.Kekulize(m)
        m = _tartrate_flattened(m)
        Chem.SanitizeMol(m)
        if not optimize_geometry:
            return m
        m = Chem.AddHs(m, addCoords=True)
        props = AllChem.MMFFGetMoleculeProperties(m, mmffVariant="MMFF94")
        ff = AllChem.MMFFGetMoleculeForceField(m, props, confId=0)
        for atom in m.GetAtoms():                     # position restraints on every atom (standardizer.py:72-74)
            ff.MMFFAddPositionConstraint(atom.GetIdx(), 0.2, 800.0)
        ff.Initialize()
        ff.Minimize(maxIts=1000, energyTol=1e-08)
        return Chem.RemoveHs(m)
    except Exception:  # noqa: BLE001 - the reference uses a bare except here
        return None


def finish_with_rdkit(gen, x, h, node_mask, optimise_geometry: bool) -> List["Chem.Mol"]:
    """conformer_generator.py:342-368 with the GCN on the HIP path."""
    mols = samples_to_rdkit_mol(positions=x, one_hot=h, node_mask=node_mask, atom_decoder=gen.atom_decoder)
    el, dm, am, ordered = prepare_adj_mat_seer_input(mols, gen.dimension, gen.device)
    if not ordered:
        return []
    bond = gen.adj_mat_seer.bond_orders(el, dm, am).cpu()
    gen.last_batch = dict(x=x, h=h, elements=el, bond=bond)
    out = []
    for k, mol in enumerate(ordered):
        std = standardize_mol(redefine_bonds(mol, bond[k]), optimize_geometry=optimise_geometry)
        if std:
            out.append(std)
    return out
