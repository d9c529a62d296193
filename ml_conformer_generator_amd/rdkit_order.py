"""Atom order and 1-order connectivity BEFORE the GCN, as the reference takes them from RDKit.

The reference renumbers every generated molecule into canonical-SMILES atom order and perceives a 1-order
connectivity with RDKit before `AdjMatSeer` sees it (`canonicalise`, utils/mol_utils.py:110-126, called from
`prepare_adj_mat_seer_input`, :146-194):

    rdDetermineBonds.DetermineConnectivity(mol); Chem.MolToSmiles(mol)
    order = mol.GetProp("_smilesAtomOutputOrder"); mol = Chem.RenumberAtoms(mol, order)
    adjacency = MolGraph.from_mol(mol).adjacency_matrix()            # > 0  -> {0,1}, + I

AdjMatSeer is NOT permutation-equivariant (`nodes_coord_fc` is a dense layer over the position index,
adj_mat_seer.py:135-138), so a checkpoint trained on that order predicts different bonds on any other.  The two decisions
are therefore inputs of the device hand-off (`mcg_handoff_ex`: `order`, `conn_in`), supplied by an *atom-order provider*:

    provider(atomic_numbers: List[int], coords: numpy.ndarray[n,3] float64) -> (order, connectivity) | None
        order         sequence of n ints: position p of the GCN input holds generation atom order[p]
                      (the argument of `Chem.RenumberAtoms`); None = keep generation order
        connectivity  [n,n] {0,1} array in GENERATION order (before the renumbering); None = covalent-radius rule
        None          the molecule could not be built (the reference drops it: `MolFromXYZBlock` returned None,
                      mol_utils.py:53-55)

`rdkit_provider` is the reference's sequence and is chosen automatically where RDKit imports; it is UNTESTED OFFLINE
(RDKit exists neither in the build container nor on the GPU boxes) - the plumbing around it (batching, validation,
upload, permuted outputs) is tested with injected fake providers, and the kernel against reference-generated fixtures
with a fixed injected order (`tests/golden/e2e_perm_T20_b4n19.npz`).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .config import ATOM_DECODER, ATOMIC_NUMBERS

Provider = Callable[[List[int], np.ndarray], Optional[Tuple[Optional[Sequence[int]], Optional[np.ndarray]]]]

_Z2SYMBOL = {z: ATOM_DECODER[i] for i, z in enumerate(ATOMIC_NUMBERS)}


def have_rdkit() -> bool:
    try:
        import rdkit  # noqa: F401
        return True
    except Exception:  # noqa: BLE001
        return False


def xyz_block(atomic_numbers: Sequence[int], coords) -> str:
    """The XYZ text `samples_to_rdkit_mol` writes (mol_utils.py:39-51: count, empty line, "%s %.9f %.9f %.9f")."""
    lines = ["%d\n\n" % len(atomic_numbers)]
    for z, c in zip(atomic_numbers, coords):
        lines.append("%s %.9f %.9f %.9f\n" % (_Z2SYMBOL[int(z)], float(c[0]), float(c[1]), float(c[2])))
    return "".join(lines)


def parse_smiles_output_order(prop: str) -> List[int]:
    """`_smilesAtomOutputOrder` ("[3,0,1,2,]") -> [3, 0, 1, 2] (mol_utils.py:119-122)."""
    prop = prop.replace("[", "").replace("]", "")
    return [int(v) for v in prop.split(",") if v != ""]


def rdkit_provider(atomic_numbers: List[int], coords: np.ndarray):
    """The reference's `samples_to_rdkit_mol` + `canonicalise` for ONE molecule (mol_utils.py:39-55,110-126).  A molecule
    without a single perceived bond raises ValueError like the reference (`MolGraph.from_mol`, molgraph.py:150-153)."""
    from rdkit import Chem
    from rdkit.Chem import rdDetermineBonds
    mol = Chem.MolFromXYZBlock(xyz_block(atomic_numbers, coords))
    if mol is None:
        return None
    rdDetermineBonds.DetermineConnectivity(mol)
    _ = Chem.MolToSmiles(mol)
    order = parse_smiles_output_order(mol.GetProp("_smilesAtomOutputOrder"))
    if mol.GetNumBonds() == 0:
        raise ValueError("Bonds must be specified for the molecule - no connectivity perceived.")
    conn = np.asarray(Chem.GetAdjacencyMatrix(mol)) != 0            # generation order: read before any renumbering
    return order, conn.astype(np.uint8)


def default_provider() -> Optional[Provider]:
    """RDKit's canonical order where RDKit imports (the reference's behaviour), else None (generation order +
    covalent-radius connectivity: the labelled substitutes of `handoff.py`)."""
    return rdkit_provider if have_rdkit() else None


def batch_order_and_connectivity(provider: Provider, x: torch.Tensor, h: torch.Tensor, n_nodes: torch.Tensor):
    """Run `provider` over a generated batch (x[B,N,3], h[B,N,8] one-hot, n_nodes[B]; any device - ONE D2H copy each).
    Returns (order, connectivity, built): per-molecule lists for `prepare_adj_mat_seer_input_hip` (None entries = the
    substitutes) and built[B] bool (False where the provider returned None: the reference drops that molecule)."""
    xc = x.detach().to("cpu", torch.float64).numpy()
    cls = torch.argmax(h.detach(), dim=2).to("cpu").tolist()          # argmax(one_hot) (mol_utils.py:41)
    ns = [int(v) for v in n_nodes.detach().to("cpu").reshape(-1).tolist()]
    order: List[Optional[Sequence[int]]] = []
    conn: List[Optional[np.ndarray]] = []
    built: List[bool] = []
    for b, n in enumerate(ns):
        z = [ATOMIC_NUMBERS[k] for k in cls[b][:n]]
        res = provider(z, xc[b, :n]) if n > 0 else (None, None)
        if res is None:
            order.append(None); conn.append(None); built.append(False)
            continue
        o, c = res
        order.append(None if o is None else [int(v) for v in o])
        conn.append(None if c is None else np.asarray(c))
        built.append(True)
    given = [c is not None for c, ok in zip(conn, built) if ok]
    if any(given) and not all(given):
        raise ValueError("an atom-order provider must return a connectivity for every molecule or for none "
                         "(the hand-off kernel applies one connectivity rule per launch)")
    if any(given):          # molecules the provider could not build are dropped downstream: any placeholder will do
        conn_arg = [np.zeros((n, n), dtype=np.uint8) if c is None else c for c, n in zip(conn, ns)]
    else:
        conn_arg = None
    order_arg = None if all(o is None for o in order) else order
    return order_arg, conn_arg, built
