"""EDM -> GCN hand-off and bond write-back on the device.

The reference goes through RDKit here (`samples_to_rdkit_mol`, `canonicalise`,
`prepare_adj_mat_seer_input`, `redefine_bonds`, `standardize_mol`;
mol_utils.py:18-57,110-223, standardizer.py).  The tensor halves (atom decode, distances + I,
connectivity + I, pad 42, lower-triangle bond write-back) run in two HIP launches
(`mcg_handoff_ex`, `mcg_bond_writeback`), follow the reference and are checked against
`oracle/host_oracle.py` and reference-generated fixtures; one D2H copy at the end.
The two decisions RDKit OWNS in front of the GCN - the canonical-SMILES atom order and the
1-order connectivity (`canonicalise`, mol_utils.py:110-126) - are INPUTS of the hand-off launch
(`order`, `connectivity`): RDKit's own where RDKit imports (`rdkit_order.py`), a caller's, or -
when neither is given - the labelled substitutes (generation order, covalent-radius rule:
parity unpinned against RDKit).  "valid" without RDKit is a valence / connectivity proxy.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from .config import ATOM_DECODER, ATOMIC_NUMBERS, DIMENSION

_COV_FACTOR = 1.3          # RDKit DetermineConnectivity's default covFactor (radii: csrc/mcg_gcn.hip k_handoff)


class _HostBatch:
    """The gathered result tensors of one generation call on the HOST (one D2H copy per tensor); the molecule records
    below are views into it."""
    __slots__ = ("x", "elements", "bonds", "n")

    def __init__(self, x, elements, bonds, n):
        self.x, self.elements, self.bonds, self.n = x, elements, bonds, n


class GeneratedMolecule:
    """RDKit-free result record: heavy atoms, coordinates and bond orders.

    `atomic_numbers` List[int]; `coords` [n,3] float32 (CPU); `bond_orders` [n,n] int8, symmetric, 0 none / 1 / 2 / 3 /
    4 aromatic; `valid` bool.  Records built by `molecules_from_tensors` are LAZY views into the batch's host tensors: the
    per-molecule slices are taken on first access, so assembling a batch costs one object per molecule (2 048 molecules:
    a few ms instead of 70-170 ms of slicing + cloning on every rank of a sharded call)."""
    __slots__ = ("_z", "_coords", "_bonds", "valid", "_batch", "_b")

    def __init__(self, atomic_numbers: List[int], coords: torch.Tensor, bond_orders: torch.Tensor, valid: bool = True):
        self._z, self._coords, self._bonds, self.valid = list(atomic_numbers), coords, bond_orders, bool(valid)
        self._batch, self._b = None, -1

    @classmethod
    def _view(cls, batch: _HostBatch, b: int, valid: bool) -> "GeneratedMolecule":
        m = cls.__new__(cls)
        m._z = m._coords = m._bonds = None
        m.valid, m._batch, m._b = valid, batch, b
        return m

    @property
    def atomic_numbers(self) -> List[int]:
        if self._z is None:
            self._z = self._batch.elements[self._b, : self._batch.n[self._b]].tolist()
        return self._z

    @property
    def coords(self) -> torch.Tensor:
        if self._coords is None:
            self._coords = self._batch.x[self._b, : self._batch.n[self._b]]
        return self._coords

    @property
    def bond_orders(self) -> torch.Tensor:
        if self._bonds is None:
            n = self._batch.n[self._b]
            self._bonds = self._batch.bonds[self._b, :n, :n]
        return self._bonds

    def __repr__(self) -> str:
        return f"GeneratedMolecule(n_atoms={self.GetNumAtoms()}, valid={self.valid})"

    def __eq__(self, other) -> bool:
        """Value equality (atoms, coordinates, bond orders, validity) - whether or not the records are views of a batch."""
        if not isinstance(other, GeneratedMolecule):
            return NotImplemented
        return (self.valid == other.valid and self.atomic_numbers == other.atomic_numbers
                and torch.equal(self.coords, other.coords) and torch.equal(self.bond_orders, other.bond_orders))

    __hash__ = None            # mutable value object, like the dataclass it replaced

    @property
    def symbols(self) -> List[str]:
        z2s = {z: ATOM_DECODER[i] for i, z in enumerate(ATOMIC_NUMBERS)}
        return [z2s[z] for z in self.atomic_numbers]

    def GetNumAtoms(self) -> int:          # RDKit-like convenience
        return len(self._z) if self._z is not None else int(self._batch.n[self._b])

    def to_molblock(self, name: str = "generated") -> str:
        """V2000 MOL block (heavy atoms, bond orders 1/2/3, aromatic = 4) - consumable by RDKit
        (`Chem.MolFromMolBlock`) or any other toolkit without going through this package."""
        n = len(self.atomic_numbers)
        bonds = [(i + 1, j + 1, int(self.bond_orders[i, j])) for i in range(n) for j in range(i)
                 if int(self.bond_orders[i, j]) != 0]
        lines = [name, "  mlconfgen-mi355x", "", "%3d%3d  0  0  0  0  0  0  0  0999 V2000" % (n, len(bonds))]
        for s, c in zip(self.symbols, self.coords.tolist()):
            lines.append("%10.4f%10.4f%10.4f %-3s 0  0  0  0  0  0  0  0  0  0  0  0" % (c[0], c[1], c[2], s))
        for i, j, o in bonds:
            lines.append("%3d%3d%3d  0" % (i, j, o))
        lines.append("M  END")
        return "\n".join(lines) + "\n"

    def to_xyz_block(self) -> str:
        lines = [f"{len(self.atomic_numbers)}", ""]
        for s, c in zip(self.symbols, self.coords.tolist()):
            lines.append("%s %.9f %.9f %.9f" % (s, c[0], c[1], c[2]))
        return "\n".join(lines) + "\n"


def _order_tensor(order, n_host: List[int], dimension: int) -> torch.Tensor:
    """[B,42] int32 (CPU) from per-molecule permutations, CHECKED: row b's first n_b entries must be a permutation of
    0..n_b-1 (the argument of `Chem.RenumberAtoms`, mol_utils.py:124); None rows = generation order."""
    if torch.is_tensor(order):
        rows = order.detach().to("cpu", torch.int64).reshape(len(n_host), -1).tolist()
    else:
        rows = [None if r is None else [int(v) for v in r] for r in order]
    if len(rows) != len(n_host):
        raise ValueError(f"order has {len(rows)} rows for a batch of {len(n_host)} molecules")
    out = torch.arange(dimension, dtype=torch.int32).repeat(len(n_host), 1)
    for b, (r, n) in enumerate(zip(rows, n_host)):
        if r is None:
            continue
        if len(r) < n or sorted(r[:n]) != list(range(n)):
            raise ValueError(f"order[{b}] is not a permutation of the molecule's {n} atoms")
        out[b, :n] = torch.tensor(r[:n], dtype=torch.int32)
    return out


def _conn_tensor(conn, n_host: List[int], dimension: int) -> torch.Tensor:
    """[B,42,42] uint8 (CPU) from per-molecule [n,n] (or padded) {0,1} connectivities in generation order."""
    if torch.is_tensor(conn) and conn.dim() == 3 and tuple(conn.shape[1:]) == (dimension, dimension):
        return (conn.detach().to("cpu") != 0).to(torch.uint8).contiguous()
    if len(conn) != len(n_host):
        raise ValueError(f"connectivity has {len(conn)} entries for a batch of {len(n_host)} molecules")
    out = torch.zeros(len(n_host), dimension, dimension, dtype=torch.uint8)
    for b, (c, n) in enumerate(zip(conn, n_host)):
        c = torch.as_tensor(c)
        if c.dim() != 2 or c.shape[0] < n or c.shape[1] < n:
            raise ValueError(f"connectivity[{b}] must be at least [{n},{n}]")
        c = c[:n, :n] != 0
        if not bool(torch.equal(c, c.t())):
            raise ValueError(f"connectivity[{b}] is not symmetric")
        out[b, :n, :n] = c.to(torch.uint8)
    return out


def prepare_adj_mat_seer_input_hip(x: torch.Tensor, h: torch.Tensor, n_nodes: torch.Tensor,
                                   dimension: int = DIMENSION, order=None, connectivity=None, with_coords: bool = False):
    """elements[B,42] i64, dist_mat[B,42,42] (+I, zero padded), adj_mat[B,42,42] {0,1} (+I) - the tensors
    `prepare_adj_mat_seer_input` (mol_utils.py:146-194) builds, by ONE HIP launch (`mcg_handoff_ex`).
    `order` (per-molecule permutations: position p holds generation atom order[b][p] - the reference's canonical SMILES
    order, `canonicalise` :110-126) and `connectivity` (per-molecule {0,1} matrices in GENERATION order - RDKit's
    `DetermineConnectivity`, :117) are the two RDKit-owned decisions; None = generation order / covalent-radius rule.
    `with_coords`: also return x[B,N,3] in the order of the GCN input (the reference's `canonicalised_samples`)."""
    from . import _lib
    if dimension != DIMENSION:
        raise ValueError("the hand-off kernel is specialised for DIMENSION = 42")
    B, N, _ = x.shape
    dev = x.device
    el = torch.empty(B, dimension, dtype=torch.long, device=dev)
    dm = torch.empty(B, dimension, dimension, dtype=torch.float32, device=dev)
    am = torch.empty(B, dimension, dimension, dtype=torch.float32, device=dev)
    xo = torch.empty(B, N, 3, dtype=torch.float32, device=dev) if with_coords else None
    if B == 0:
        return (el, dm, am, xo) if with_coords else (el, dm, am)
    nn = n_nodes.to(dev, torch.int32).contiguous()
    od = cd = None
    if order is not None or connectivity is not None:
        n_host = [int(v) for v in n_nodes.detach().to("cpu").reshape(-1).tolist()]
        if order is not None:
            od = _order_tensor(order, n_host, dimension).to(dev)
        if connectivity is not None:
            cd = _conn_tensor(connectivity, n_host, dimension).to(dev)
    _lib.check(_lib.lib().mcg_handoff_ex(_lib.dptr(x.contiguous()), _lib.dptr(h.to(torch.float32).contiguous()), _lib.dptr(nn),
                                         B, N, _COV_FACTOR, _lib.dptr(od), _lib.dptr(cd), _lib.dptr(el), _lib.dptr(dm),
                                         _lib.dptr(am), _lib.dptr(xo), None, _lib.current_stream_ptr(dev)), "mcg_handoff_ex")
    return (el, dm, am, xo) if with_coords else (el, dm, am)


def bond_writeback_hip(bond: torch.Tensor, elements: torch.Tensor, n_nodes: torch.Tensor
                       ) -> Tuple[torch.Tensor, torch.Tensor]:
    """(bond_sym [B,42,42] int8, valid_proxy [B] bool) by ONE HIP launch (`mcg_bond_writeback`): the strict lower
    triangle of the GCN's bond argmax mirrored (`redefine_bonds`, mol_utils.py:210-211) and the valence /
    single-fragment pre-filter that stands in for `standardize_mol(...) is not None` (a labelled proxy)."""
    from . import _lib
    B = int(bond.shape[0])
    dev = bond.device
    sym = torch.empty(B, DIMENSION, DIMENSION, dtype=torch.int8, device=dev)
    valid = torch.empty(B, dtype=torch.uint8, device=dev)
    if B == 0:
        return sym, valid.bool()
    nn = n_nodes.to(dev, torch.int32).contiguous()
    _lib.check(_lib.lib().mcg_bond_writeback(_lib.dptr(bond.to(torch.int8).contiguous()), _lib.dptr(elements.contiguous()),
                                             _lib.dptr(nn), B, _lib.dptr(sym), _lib.dptr(valid),
                                             _lib.current_stream_ptr(dev)), "mcg_bond_writeback")
    return sym, valid.bool()


def molecules_from_tensors(x: torch.Tensor, elements: torch.Tensor, bond_sym: torch.Tensor, n_nodes: torch.Tensor,
                           valid: torch.Tensor) -> List[GeneratedMolecule]:
    """Host pass over (already gathered) result tensors: ONE D2H copy per tensor, then one lazy record per molecule
    (no per-molecule slicing or cloning here: `GeneratedMolecule._view`)."""
    batch = _HostBatch(x.cpu(), elements.cpu(), bond_sym.cpu(), n_nodes.cpu().tolist())
    view = GeneratedMolecule._view
    return [view(batch, b, bool(v)) for b, v in enumerate(valid.cpu().tolist())]


def assemble_molecules(x: torch.Tensor, elements: torch.Tensor, bond: torch.Tensor, n_nodes: torch.Tensor
                       ) -> List[GeneratedMolecule]:
    """Bond write-back + validity proxy on the device (one launch), ONE D2H copy of (x, elements, bonds, valid),
    then a slicing-only host pass (the reference standardises one molecule at a time,
    conformer_generator.py:362-366)."""
    sym, valid = bond_writeback_hip(bond, elements, n_nodes)
    return molecules_from_tensors(x, elements, sym, n_nodes, valid)
