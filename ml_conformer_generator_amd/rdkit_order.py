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

from . import _rdkit_tasks as _tasks
from . import host_pool
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
    return _tasks.xyz_text(atomic_numbers, coords, _Z2SYMBOL)


parse_smiles_output_order = _tasks.parse_smiles_output_order


def rdkit_provider(atomic_numbers: List[int], coords: np.ndarray):
    """The reference's `samples_to_rdkit_mol` + `canonicalise` for ONE molecule (mol_utils.py:39-55,110-126).  A molecule
    without a single perceived bond raises ValueError like the reference (`MolGraph.from_mol`, molgraph.py:150-153).
    The body lives in `_rdkit_tasks.order_one` (the file the pool's workers load); this is the in-process form."""
    return _tasks.order_one(atomic_numbers, coords, _Z2SYMBOL)


def default_provider() -> Optional[Provider]:
    """RDKit's canonical order where RDKit imports (the reference's behaviour), else None (generation order +
    covalent-radius connectivity: the labelled substitutes of `handoff.py`)."""
    return rdkit_provider if have_rdkit() else None


def provider_task(provider):
    """(TaskRef, args) when `provider` can run in the host pool's workers - RDKit's own sequence, or a caller's chunk
    function named by a `host_pool.TaskRef` (`f(items) -> list`, item = (atomic_numbers, coords float64 [n,3]), result =
    what a provider returns) - else None: a plain callable runs in this process, one molecule at a time."""
    if provider is rdkit_provider:
        return host_pool.ORDER_TASK, (_Z2SYMBOL,)
    if isinstance(provider, host_pool.TaskRef):
        return provider, ()
    return None


def launch_groups(batch: int, max_groups: int = 4, min_group: int = 8) -> List[Tuple[int, int]]:
    """Contiguous molecule groups [lo, hi) of one generated batch: the hand-off + GCN of a group is launched as soon as
    ITS order results are in, while the host is still working on the later groups."""
    n_groups = max(1, min(max_groups, batch // min_group))
    base, extra = divmod(batch, n_groups)
    out, lo = [], 0
    for g in range(n_groups):
        hi = lo + base + (1 if g < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


class OrderStage:
    """The order + connectivity decisions of one generated batch, computed per launch group.

    Construction copies x, h, n_nodes to the host (ONE D2H copy each) and - for a poolable provider - submits every
    task right away; `result(g)` blocks until group g's molecules are done and returns what
    `prepare_adj_mat_seer_input_hip` takes for x[lo:hi]: (order, connectivity, built)."""

    def __init__(self, provider, x: torch.Tensor, h: torch.Tensor, n_nodes: torch.Tensor, executor=None,
                 groups: Optional[List[Tuple[int, int]]] = None):
        xc = x.detach().to("cpu", torch.float64).numpy()
        cls = torch.argmax(h.detach(), dim=2).to("cpu").tolist()          # argmax(one_hot) (mol_utils.py:41)
        self.ns = [int(v) for v in n_nodes.detach().to("cpu").reshape(-1).tolist()]
        self.items = [([ATOMIC_NUMBERS[k] for k in cls[b][:n]], xc[b, :n]) for b, n in enumerate(self.ns)]
        self.groups = groups if groups is not None else [(0, len(self.ns))]
        self.provider = provider
        self._given = None                     # did the provider supply connectivities?  one answer per batch
        task = provider_task(provider)
        self._futs = None
        if task is not None:
            ref, args = task
            ex = executor if executor is not None else host_pool.SerialExecutor()
            chunk = host_pool.task_chunk(len(self.items), getattr(ex, "n_workers", 0))
            self._futs = []
            for lo, hi in self.groups:
                self._futs.append([ex.submit(ref, self.items[a:b], args) for a, b in
                                   ((lo + c0, lo + c1) for c0, c1 in host_pool.chunk_bounds(hi - lo, chunk))])

    def _raw(self, g: int) -> list:
        lo, hi = self.groups[g]
        if self._futs is not None:
            out: list = []
            for f in self._futs[g]:
                out.extend(f.result())
            if len(out) != hi - lo:
                raise ValueError(f"the order task returned {len(out)} results for {hi - lo} molecules")
            return out
        return [self.provider(z, c) if len(z) > 0 else None for z, c in self.items[lo:hi]]      # empty = not buildable = dropped

    def result(self, g: int):
        lo, hi = self.groups[g]
        ns = self.ns[lo:hi]
        order: List[Optional[Sequence[int]]] = []
        conn: List[Optional[np.ndarray]] = []
        built: List[bool] = []
        for res in self._raw(g):
            if res is None:
                order.append(None); conn.append(None); built.append(False)
                continue
            o, c = res
            order.append(None if o is None else [int(v) for v in o])
            conn.append(None if c is None else np.asarray(c))
            built.append(True)
        given = [c is not None for c, ok in zip(conn, built) if ok]
        mixed = any(given) and not all(given)
        if given and not mixed:
            if self._given is None:
                self._given = given[0]
            mixed = self._given != given[0]
        if mixed:
            raise ValueError("an atom-order provider must return a connectivity for every molecule or for none "
                             "(the hand-off kernel applies one connectivity rule per launch)")
        use_conn = any(given) if given else bool(self._given)
        if use_conn:          # molecules the provider could not build are dropped downstream: any placeholder will do
            conn_arg = [np.zeros((n, n), dtype=np.uint8) if c is None else c for c, n in zip(conn, ns)]
        else:
            conn_arg = None
        order_arg = None if all(o is None for o in order) else order
        return order_arg, conn_arg, built


def batch_order_and_connectivity(provider: Provider, x: torch.Tensor, h: torch.Tensor, n_nodes: torch.Tensor,
                                 executor=None):
    """Run `provider` over a generated batch (x[B,N,3], h[B,N,8] one-hot, n_nodes[B]; any device - ONE D2H copy each).
    Returns (order, connectivity, built): per-molecule lists for `prepare_adj_mat_seer_input_hip` (None entries = the
    substitutes) and built[B] bool (False where the provider returned None: the reference drops that molecule).
    `executor`: a `host_pool.HostPool` fans a poolable provider (`provider_task`) out over host cores."""
    return OrderStage(provider, x, h, n_nodes, executor).result(0)
