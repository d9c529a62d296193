"""Optional RDKit finish of `generate_conformers` (conformer_generator.py:362-366 -> utils/standardizer.py:83-111).

Runs ONLY where RDKit imports.  RDKit is absent from the build container and from the GPU boxes of this project, so this
module is UNTESTED OFFLINE and its parity with the reference is unpinned; the hot path never
depends on it.  It restates the reference's validity gate on the molecules the HIP path produced:

    largest fragment -> Kekulize -> drop the stereo tags of free tartrate fragments -> SanitizeMol ->
    [optimise_geometry: AddHs(addCoords) -> MMFF94 minimisation with every atom position-restrained (0.2 A, 800
     kcal/mol/A^2, <= 1000 iterations, energyTol 1e-8) -> RemoveHs]                      any exception => dropped

Input: `GeneratedMolecule` records (atoms in generation order, bond orders from the GCN's lower-triangle argmax -
`mcg_bond_writeback`), handed over as V2000 mol blocks.  The reference instead perceives connectivity with RDKit and
reorders atoms canonically before the GCN (mol_utils.py:110-194): those two RDKit decisions stay substituted (handoff.py).
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
    """`GeneratedMolecule` records -> RDKit Mols through the reference's gate; None where it rejects one."""
    from rdkit import Chem
    out = []
    for rec in molecules:
        mol = Chem.MolFromMolBlock(rec.to_molblock(), sanitize=False, removeHs=False)
        out.append(None if mol is None else _standardize(mol, optimise_geometry))
    return out
