"""Optional RDKit finish of `generate_conformers` (conformer_generator.py:357-366 -> utils/mol_utils.py:197-223 ->
utils/standardizer.py:83-111).

Runs ONLY where RDKit imports.  RDKit is absent from the build container and from the GPU boxes of this project, so this
module is UNTESTED AGAINST RDKIT OFFLINE and its parity with the reference is unpinned; the hot
path never depends on it.  What IS tested offline is its call sequence, against a recording stand-in for `rdkit.Chem`
(`tests/test_host_logic.py::test_rdkit_finish_follows_the_reference_call_sequence`).  `tests/test_rdkit_optional.py`
(`pytest.importorskip("rdkit")`) checks it against real RDKit wherever that exists.

The Mol is built the way the reference builds it, NOT through a mol block (a V2000 block rounds coordinates to 1e-4 A, marks
type-4 bonds and their atoms aromatic, and lets the parser perceive chirality from the conformer - none of which the
reference's route does):

    redefine_bonds (mol_utils.py:197-223):
        mol   = the canonicalised sample = MolFromXYZBlock("%.9f" text of the generated atoms)      (:39-53)
        c_mol = MolFromXYZBlock(MolToXYZBlock(mol))             bonds and atom properties stripped  (:206-207)
        for i in range(n): for j in range(n): bond_type = tril(argmax)[i, j]; != 0 -> AddBond(i, j, bond_type_dict[...])
    standardize_mol (standardizer.py:83-111):
        largest fragment -> Kekulize -> drop the stereo tags of free tartrate fragments -> SanitizeMol ->
        [optimise_geometry: AddHs(addCoords) -> MMFF94 minimisation with every atom position-restrained (0.2 A, 800
         kcal/mol/A^2, <= 1000 iterations, energyTol 1e-8) -> RemoveHs]                  any exception => dropped

Input: `GeneratedMolecule` records (atoms in the order the GCN saw them - RDKit's canonical order where
`rdkit_order.rdkit_provider` ran - bond orders from the GCN's lower-triangle argmax, `mcg_bond_writeback`).
"""
from __future__ import annotations

from typing import List, Optional

_TARTRATE_SMARTS = "OC(=O)C(O)C(O)C(=O)O"


def have_rdkit() -> bool:
    try:
        import rdkit  # noqa: F401
        return True
    except Exception:  # noqa: BLE001
        return False


def _bond_type_dict(Chem):
    bt = Chem.rdchem.BondType                                            # mol_utils.py:10-15
    return {1: bt.SINGLE, 2: bt.DOUBLE, 3: bt.TRIPLE, 4: bt.AROMATIC}


def mol_from_record(rec, Chem=None):
    """`redefine_bonds` (mol_utils.py:197-223) on a `GeneratedMolecule`: XYZ text -> Mol -> XYZ text -> Mol (the
    reference's two text round trips: "%.9f", then MolToXYZBlock's own precision), then one AddBond per non-zero entry of
    the strict lower triangle, in the reference's (i, j) loop order."""
    if Chem is None:
        from rdkit import Chem
    mol = Chem.MolFromXYZBlock(rec.to_xyz_block())
    if mol is None:
        return None
    c_mol = Chem.MolFromXYZBlock(Chem.MolToXYZBlock(mol))
    ed_mol = Chem.EditableMol(c_mol)
    types = _bond_type_dict(Chem)
    bo = rec.bond_orders.tolist()
    n = len(bo)
    for i in range(n):
        for j in range(i):                       # tril with the diagonal removed (:210-211): only j < i can be non-zero
            t = int(bo[i][j])
            if t != 0:
                ed_mol.AddBond(i, j, types[t])
    return ed_mol.GetMol()


def mol_without_bonds(rec, Chem=None):
    """`samples_to_rdkit_mol` for one record (mol_utils.py:39-55): the Mol `edm_samples` returns (no bonds)."""
    if Chem is None:
        from rdkit import Chem
    return Chem.MolFromXYZBlock(rec.to_xyz_block())


def _standardize(mol, optimise_geometry: bool):
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


def finish(molecules: List, optimise_geometry: bool = True) -> List[Optional[object]]:
    """`GeneratedMolecule` records -> RDKit Mols through the reference's `redefine_bonds` + `standardize_mol`; None where
    the gate rejects one (conformer_generator.py:362-366 drops those)."""
    out = []
    for rec in molecules:
        mol = mol_from_record(rec)
        out.append(None if mol is None else _standardize(mol, optimise_geometry))
    return out


def samples(molecules: List) -> List[object]:
    """`edm_samples`' return value (conformer_generator.py:262-266): bond-free Mols; unbuildable ones are skipped
    (mol_utils.py:53-55)."""
    out = []
    for rec in molecules:
        mol = mol_without_bonds(rec)
        if mol is not None:
            out.append(mol)
    return out
