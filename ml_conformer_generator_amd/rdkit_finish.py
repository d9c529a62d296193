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

from typing import List, Optional, Sequence, Tuple

from . import _rdkit_tasks as _tasks
from . import host_pool
from .config import ATOM_DECODER, ATOMIC_NUMBERS

_Z2SYMBOL = {z: ATOM_DECODER[i] for i, z in enumerate(ATOMIC_NUMBERS)}


def have_rdkit() -> bool:
    try:
        import rdkit  # noqa: F401
        return True
    except Exception:  # noqa: BLE001
        return False


def mol_from_record(rec, Chem=None):
    """`redefine_bonds` (mol_utils.py:197-223) on a `GeneratedMolecule`: XYZ text -> Mol -> XYZ text -> Mol (the
    reference's two text round trips: "%.9f", then MolToXYZBlock's own precision), then one AddBond per non-zero entry of
    the strict lower triangle, in the reference's (i, j) loop order (body: `_rdkit_tasks.mol_from_xyz_and_bonds`)."""
    return _tasks.mol_from_xyz_and_bonds(rec.to_xyz_block(), rec.bond_orders.tolist(), Chem)


def mol_without_bonds(rec, Chem=None):
    """`samples_to_rdkit_mol` for one record (mol_utils.py:39-55): the Mol `edm_samples` returns (no bonds)."""
    if Chem is None:
        from rdkit import Chem
    return Chem.MolFromXYZBlock(rec.to_xyz_block())


_standardize = _tasks.standardize


def record_item(rec) -> Tuple[list, list, list]:
    """A `GeneratedMolecule` as the plain-data item the finish tasks take (atomic numbers, coordinates as python floats
    holding the fp32 values - what "%.9f" prints -, bond-order rows)."""
    return rec.atomic_numbers, rec.coords.tolist(), rec.bond_orders.tolist()


def finisher_task(finisher, optimise_geometry: bool):
    """(TaskRef, args, rebuild) for a finisher the host pool can run: "rdkit" = the reference's `redefine_bonds` +
    `standardize_mol` (results travel as `Mol.ToBinary()` bytes and are rebuilt here), or a caller's chunk function named
    by a `host_pool.TaskRef` (`f(items, optimise_geometry) -> list`, results returned as they are; None = dropped)."""
    if finisher == "rdkit":
        def rebuild(b):
            from rdkit import Chem
            return None if b is None else Chem.Mol(b)
        return host_pool.FINISH_TASK, (bool(optimise_geometry), _Z2SYMBOL), rebuild
    if isinstance(finisher, host_pool.TaskRef):
        return finisher, (bool(optimise_geometry),), (lambda r: r)
    raise ValueError(f"unknown finisher {finisher!r}")


class FinishStage:
    """The RDKit finish of one generated batch, submitted group by group (`add`) and collected in sample order
    (`results`).  With a `SerialExecutor` and the RDKit finisher the Mols are built in-process (no byte round trip)."""

    def __init__(self, finisher, optimise_geometry: bool, executor=None):
        self.executor = executor if executor is not None else host_pool.SerialExecutor()
        self.serial_rdkit = finisher == "rdkit" and getattr(self.executor, "n_workers", 0) == 0
        self.optimise_geometry = bool(optimise_geometry)
        if not self.serial_rdkit:
            self.ref, self.args, self.rebuild = finisher_task(finisher, optimise_geometry)
        self._parts: List = []                 # per group: list of futures (pooled) or a list of results (serial RDKit)
        self.n_submitted = 0

    def add(self, records: Sequence) -> None:
        """Submit the finish of these molecules (sample order continues where the previous `add` stopped)."""
        items = [record_item(r) for r in records]
        self.n_submitted += len(items)
        if self.serial_rdkit:
            self._parts.append([_tasks.finish_one(z, c, bo, self.optimise_geometry, _Z2SYMBOL) for z, c, bo in items])
            return
        chunk = host_pool.task_chunk(len(items), getattr(self.executor, "n_workers", 0), cap=4)
        self._parts.append([self.executor.submit(self.ref, items[lo:hi], self.args)
                            for lo, hi in host_pool.chunk_bounds(len(items), chunk)])

    def results(self) -> List[Optional[object]]:
        out: List[Optional[object]] = []
        for part in self._parts:
            if self.serial_rdkit:
                out.extend(part)
                continue
            for f in part:
                out.extend(self.rebuild(r) for r in f.result())
        if len(out) != self.n_submitted:
            raise ValueError(f"the finish task returned {len(out)} results for {self.n_submitted} molecules")
        return out


def finish(molecules: List, optimise_geometry: bool = True, executor=None) -> List[Optional[object]]:
    """`GeneratedMolecule` records -> RDKit Mols through the reference's `redefine_bonds` + `standardize_mol`; None where
    the gate rejects one (conformer_generator.py:362-366 drops those).  `executor`: a `host_pool.HostPool` fans the
    molecules out over host cores (same per-molecule code, results in the same order)."""
    stage = FinishStage("rdkit", optimise_geometry, executor)
    stage.add(molecules)
    return stage.results()


def samples(molecules: List, executor=None) -> List[object]:
    """`edm_samples`' return value (conformer_generator.py:262-266): bond-free Mols; unbuildable ones are skipped
    (mol_utils.py:53-55)."""
    if executor is not None and getattr(executor, "n_workers", 0) > 0:
        from rdkit import Chem
        raw = host_pool.map_ordered(executor, host_pool.SAMPLES_TASK, [record_item(r) for r in molecules], (_Z2SYMBOL,))
        return [Chem.Mol(b) for b in raw if b is not None]
    out = []
    for rec in molecules:
        mol = mol_without_bonds(rec)
        if mol is not None:
            out.append(mol)
    return out
