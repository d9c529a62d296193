"""The RDKit-owned per-molecule work of `generate_conformers`, as plain functions over plain data.

This file imports NOTHING from the package (no torch, no `_lib`, no relative imports): the worker processes of
`host_pool.HostPool` load it BY FILE PATH, so a worker holds numpy + RDKit only - never the HIP library, never a GPU
context.  The parent process uses the very same functions for the serial path (`n_host_workers=0`) through
`rdkit_order.py` / `rdkit_finish.py`, so pooled and serial runs execute identical code per molecule.

Two stages, each a *chunk function* `f(items, *args) -> list` with one result per item:

  order_chunk   BEFORE the GCN: `samples_to_rdkit_mol` + `canonicalise` (utils/mol_utils.py:39-55,110-126)
                item   (atomic_numbers: list[int], coords: float64 ndarray [n,3])
                result None (Mol could not be built: the reference drops the molecule) | (order: list[int], conn: uint8 [n,n])
                raises ValueError for a molecule without a perceived bond (`MolGraph.from_mol`, utils/molgraph.py:152-155)
  finish_chunk  BEHIND the GCN: `redefine_bonds` + `standardize_mol` (utils/mol_utils.py:197-223, utils/standardizer.py:62-111)
                item   (atomic_numbers, coords: list of [x,y,z] python floats holding fp32 values, bond_orders: list of int rows)
                result None (the gate rejects it: conformer_generator.py:364-366 drops it) | `Mol.ToBinary()` bytes
  samples_chunk `edm_samples`' bond-free Mols (conformer_generator.py:262-266), same item, bytes | None

UNTESTED AGAINST RDKIT OFFLINE (RDKit exists neither in the build container nor on the GPU boxes): the call sequences are
restated from the reference line by line and checked against recording stand-ins; `tests/test_rdkit_optional.py` runs
them against real RDKit wherever that exists.
"""
from __future__ import annotations

_TARTRATE_SMARTS = "OC(=O)C(O)C(O)C(=O)O"


def xyz_text(atomic_numbers, coords, z2symbol) -> str:
    """The XYZ text `samples_to_rdkit_mol` writes (mol_utils.py:39-51: count, empty line, "%s %.9f %.9f %.9f")."""
    lines = ["%d\n\n" % len(atomic_numbers)]
    for z, c in zip(atomic_numbers, coords):
        lines.append("%s %.9f %.9f %.9f\n" % (z2symbol[int(z)], float(c[0]), float(c[1]), float(c[2])))
    return "".join(lines)


def parse_smiles_output_order(prop: str):
    """`_smilesAtomOutputOrder` ("[3,0,1,2,]") -> [3, 0, 1, 2] (mol_utils.py:119-122)."""
    prop = prop.replace("[", "").replace("]", "")
    return [int(v) for v in prop.split(",") if v != ""]


# ------------------------------------------------------------------------------------------- before the GCN
def order_one(atomic_numbers, coords, z2symbol):
    """The reference's `samples_to_rdkit_mol` + `canonicalise` for ONE molecule (mol_utils.py:39-55,110-126)."""
    import numpy as np
    from rdkit import Chem
    from rdkit.Chem import rdDetermineBonds
    mol = Chem.MolFromXYZBlock(xyz_text(atomic_numbers, coords, z2symbol))
    if mol is None:
        return None
    rdDetermineBonds.DetermineConnectivity(mol)
    _ = Chem.MolToSmiles(mol)
    order = parse_smiles_output_order(mol.GetProp("_smilesAtomOutputOrder"))
    if mol.GetNumBonds() == 0:
        raise ValueError("Bonds must be specified for the molecule - no connectivity perceived.")
    conn = np.asarray(Chem.GetAdjacencyMatrix(mol)) != 0            # generation order: read before any renumbering
    return order, conn.astype(np.uint8)


def order_chunk(items, z2symbol):
    # an EMPTY molecule cannot be built (`MolFromXYZBlock` of "0 atoms" is None: dropped) - not "built, no connectivity"
    return [order_one(z, c, z2symbol) if len(z) > 0 else None for z, c in items]


# ------------------------------------------------------------------------------------------- behind the GCN
def _bond_type_dict(Chem):
    bt = Chem.rdchem.BondType                                            # mol_utils.py:10-15
    return {1: bt.SINGLE, 2: bt.DOUBLE, 3: bt.TRIPLE, 4: bt.AROMATIC}


def mol_from_xyz_and_bonds(xyz: str, bond_rows, Chem=None):
    """`redefine_bonds` (mol_utils.py:197-223): XYZ text -> Mol -> XYZ text -> Mol (the reference's two text round trips:
    "%.9f", then MolToXYZBlock's own precision), then one AddBond per non-zero entry of the strict lower triangle, in the
    reference's (i, j) loop order.  NOT a mol block (a V2000 block rounds coordinates to 1e-4 A, marks type-4 bonds and
    their atoms aromatic, and lets the parser perceive chirality from the conformer)."""
    if Chem is None:
        from rdkit import Chem
    mol = Chem.MolFromXYZBlock(xyz)
    if mol is None:
        return None
    c_mol = Chem.MolFromXYZBlock(Chem.MolToXYZBlock(mol))
    ed_mol = Chem.EditableMol(c_mol)
    types = _bond_type_dict(Chem)
    n = len(bond_rows)
    for i in range(n):
        row = bond_rows[i]
        for j in range(i):                       # tril with the diagonal removed (:210-211): only j < i can be non-zero
            t = int(row[j])
            if t != 0:
                ed_mol.AddBond(i, j, types[t])
    return ed_mol.GetMol()


def standardize(mol, optimise_geometry: bool):
    """`standardize_mol` (standardizer.py:83-111): largest fragment -> Kekulize -> drop the stereo tags of free tartrate
    fragments -> SanitizeMol -> [AddHs(addCoords) -> MMFF94 with every atom position-restrained (0.2 A, 800 kcal/mol/A^2,
    <= 1000 iterations, energyTol 1e-8) -> RemoveHs]; any exception => None (the reference's bare `except:`)."""
    from rdkit import Chem
    from rdkit.Chem import AllChem
    from rdkit.Chem.MolStandardize import rdMolStandardize
    try:
        m = rdMolStandardize.FragmentParent(mol)                      # standardizer.py:92
        Chem.Kekulize(m)                                              # :94
        query = Chem.MolFromSmarts(_TARTRATE_SMARTS)                  # :47-59: free tartrate / tartaric acid only
        params = Chem.AdjustQueryParameters.NoAdjustments()
        params.adjustDegree = True
        params.adjustDegreeFlags = Chem.AdjustQueryWhichFlags.ADJUST_IGNORENONE
        hits = m.GetSubstructMatches(Chem.AdjustQueryProperties(query, params))
        if hits:
            m = Chem.Mol(m)
            for hit in hits:
                for k in (3, 5):
                    m.GetAtomWithIdx(hit[k]).SetChiralTag(Chem.ChiralType.CHI_UNSPECIFIED)
        Chem.SanitizeMol(m)                                           # :99
        if not optimise_geometry:
            return m
        m = Chem.AddHs(m, addCoords=True)                             # :102
        props = AllChem.MMFFGetMoleculeProperties(m, mmffVariant="MMFF94")      # :69-70
        ff = AllChem.MMFFGetMoleculeForceField(m, props, confId=0)
        for atom in m.GetAtoms():                                     # :73-74
            ff.MMFFAddPositionConstraint(atom.GetIdx(), 0.2, 800.0)
        ff.Initialize()
        ff.Minimize(maxIts=1000, energyTol=1e-08)                     # :77-78
        return Chem.RemoveHs(m)                                       # :104
    except Exception:  # noqa: BLE001 - the reference's bare `except:` (standardizer.py:108-109): invalid => dropped
        return None


def _to_bytes(mol):
    """A Mol as bytes for the trip back to the parent (`Chem.Mol(bytes)` rebuilds it): conformers travel always,
    properties only when asked for - ask for all of them."""
    from rdkit import Chem
    Chem.SetDefaultPickleProperties(Chem.PropertyPickleOptions.AllProps)
    return mol.ToBinary()


def finish_one(atomic_numbers, coords, bond_rows, optimise_geometry, z2symbol):
    mol = mol_from_xyz_and_bonds(xyz_text(atomic_numbers, coords, z2symbol), bond_rows)
    return None if mol is None else standardize(mol, optimise_geometry)


def finish_chunk(items, optimise_geometry, z2symbol):
    out = []
    for z, c, bo in items:
        mol = finish_one(z, c, bo, optimise_geometry, z2symbol)
        out.append(None if mol is None else _to_bytes(mol))
    return out


def samples_chunk(items, z2symbol):
    from rdkit import Chem
    out = []
    for z, c, _ in items:
        mol = Chem.MolFromXYZBlock(xyz_text(z, c, z2symbol))
        out.append(None if mol is None else _to_bytes(mol))
    return out
