"""`MLConformerGenerator` - drop-in for the reference class on the denoising hot path.

Same constructor kwargs, method names, argument order and errors as
`mlconfgen.MLConformerGenerator` (conformer_generator.py:19-399).  Differences, all
forced by the scope of this build:
  * `device` must be a ROCm GPU (default cuda:0): there is no CPU path;
  * `edm_weights` / `adj_mat_seer_weights` may also be an in-memory state dict;
  * the RDKit-owned stages BEFORE the GCN run as the native hand-off of `handoff.py` (one HIP launch).  Its two
    RDKit-owned decisions - canonical-SMILES atom order and 1-order connectivity (`canonicalise`, mol_utils.py:110-126) -
    are inputs of that launch, supplied by `atom_order_provider`: "auto" (default) = RDKit's where RDKit imports
    (`rdkit_order.rdkit_provider`, the reference's own call sequence), else the labelled substitutes (generation order,
    covalent-radius rule); a callable = the caller's own; None = the substitutes.  AdjMatSeer depends on the atom
    order, so with a trained checkpoint the substitutes predict different bonds than the reference.  The returned
    molecules carry their atoms in the order the GCN saw them, like the reference's `canonicalised_samples`;
  * RETURN TYPE: where RDKit imports, `generate_conformers` returns `List[Chem.Mol]` like the
    reference - the HIP path's molecules go through `rdkit_finish.finish` (the reference's
    `standardize_mol` sequence incl. MMFF when `optimise_geometry`; untested offline, RDKit is
    absent here); without RDKit it returns `GeneratedMolecule` records filtered by the labelled
    valence / single-fragment PROXY, and `optimise_geometry=True` cannot be honoured (one
    warning per process).  An RDKit Mol is accepted as reference conformer / fixed fragment;
  * `n_host_workers` (default min(32, this rank's share of the host cores); 0 = serial): the two RDKit stages - canonical order + connectivity
    before the GCN, `redefine_bonds` + `standardize_mol` (MMFF) behind it - are fanned out over a pool of fresh worker
    processes (`host_pool.py`) and pipelined against the GPU per group of molecules; the reference runs them one molecule
    at a time on the calling thread (conformer_generator.py:343-366).  Same per-molecule code, same results, same order;
    the workers are started in the background BEFORE the sampler is launched, and every task has a deadline
    (`task_timeout_s`, default 60 s per chunk of <= 8 molecules; None = none): a worker stuck inside RDKit is killed and
    replaced and its molecules are dropped like any other the gate rejects (utils/standardizer.py:108-109), with a warning;
  * `cpu_affinity` (default None = leave the process alone; "auto" = pin this process - torch's threads and the host
    pool's workers, which inherit the mask - to the cores of its GPU's NUMA node, shared between the ranks of the node:
    `affinity.py`, sysfs only, no HIP call; or an explicit list of cores);
  * `generate_conformers_sharded(...)`: the same call, batch-sharded over the ranks of an
    initialised `torch.distributed` group (one process per GPU, one gather at the end).
"""
from __future__ import annotations

import os
import time
import warnings
from typing import List, Optional, Union

import torch

from . import _lib
from .adj_mat_seer import AdjMatSeer
from .config import (ATOM_DECODER, CONTEXT_NORMS, DIMENSION, MAX_N_NODES, MIN_N_NODES, NOISE_PRECISION,
                     NUM_BOND_TYPES)
from .egnn import EGNNDynamics
from .equivariant_diffusion import EquivariantDiffusion, PredefinedNoiseSchedule
from . import distributed as mcg_dist
from . import host_pool
from . import rdkit_order
from .handoff import (GeneratedMolecule, assemble_molecules, bond_writeback_hip, molecules_from_tensors,
                      prepare_adj_mat_seer_input_hip)
from .mol_utils import (get_context_shape, ifm_get_xh_from_fragment, ifm_merge_hip,
                        ifm_prepare_gen_fragment_context, parse_molblock_heavy_atoms, prepare_edm_input,
                        prepare_fragment)

try:  # RDKit is optional on this path
    from rdkit import Chem  # type: ignore
    HAVE_RDKIT = True
except Exception:  # noqa: BLE001
    Chem = None
    HAVE_RDKIT = False


_WARNED_NO_MMFF = [False]


def _finish(mols: List, optimise_geometry: bool, use_rdkit: Optional[bool] = None):
    """(returned list, valid fraction): the reference's RDKit gate where RDKit exists (serial form; the generator's own
    route is the staged, pooled `rdkit_finish.FinishStage`), else the proxy filter."""
    if not mols:
        return [], 0.0
    if HAVE_RDKIT if use_rdkit is None else use_rdkit:
        from . import rdkit_finish
        done = rdkit_finish.finish(mols, optimise_geometry)
        kept = [m for m in done if m is not None]
        return kept, len(kept) / len(mols)
    if optimise_geometry and not _WARNED_NO_MMFF[0]:
        _WARNED_NO_MMFF[0] = True
        warnings.warn("ml_conformer_generator_amd: RDKit is not installed - optimise_geometry=True (MMFF94) is NOT applied and "
                      "validity is the labelled valence / single-fragment proxy; pass optimise_geometry=False to silence this",
                      RuntimeWarning, stacklevel=3)
    kept = [m for m in mols if m.valid]
    return kept, len(kept) / len(mols)


def _load_state_dict(src, device) -> dict:
    if isinstance(src, dict):
        return src["state_dict"] if "state_dict" in src else src
    return torch.load(src, map_location="cpu")["state_dict"]      # conformer_generator.py:90-102


class MLConformerGenerator(torch.nn.Module):
    def __init__(self, diffusion_steps: int = 100, device: Optional[torch.device] = None,
                 dimension: int = DIMENSION, num_bond_types: int = NUM_BOND_TYPES,
                 min_n_nodes: int = MIN_N_NODES, max_n_nodes: int = MAX_N_NODES,
                 context_norms: dict = CONTEXT_NORMS, atom_decoder: dict = ATOM_DECODER,
                 edm_weights: Union[str, dict] = "./edm_moi_chembl_15_39.pt",
                 adj_mat_seer_weights: Union[str, dict] = "./adj_mat_seer_chembl_15_39.pt",
                 compute_dtype: str = "f32", atom_order_provider="auto", n_host_workers: Optional[int] = None,
                 finisher="auto", task_timeout_s: Optional[float] = host_pool.DEFAULT_TASK_TIMEOUT_S, cpu_affinity=None):
        super().__init__()
        _lib.lib()       # fail loudly if the HIP library is not built
        device = torch.device("cuda:0" if device is None else device)
        if device.type != "cuda":
            raise ValueError("ml_conformer_generator_amd runs the hot path on an MI355X only; "
                             f"device={device} is not supported (no CPU fallback)")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        # placement first: threads and worker processes created from here on inherit it (affinity.py; sysfs only)
        self.cpu_affinity = None
        if cpu_affinity is not None:
            from . import affinity
            cpus = affinity.rank_cpus(int(os.environ.get("LOCAL_RANK", "0") or 0),
                                      int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1),
                                      device.index) if cpu_affinity == "auto" else list(cpu_affinity)
            self.cpu_affinity = affinity.pin(cpus)
        self.dimension = dimension
        self.context_norms = {k: torch.tensor(v) for k, v in context_norms.items()}
        self.atom_decoder = atom_decoder
        self.min_n_nodes, self.max_n_nodes = min_n_nodes, max_n_nodes

        net_dynamics = EGNNDynamics(in_node_nf=9, context_node_nf=3, hidden_nf=420, device=device)
        generative_model = EquivariantDiffusion(dynamics=net_dynamics, in_node_nf=8, timesteps=1000,
                                                noise_precision=NOISE_PRECISION)
        adj_mat_seer = AdjMatSeer(dimension=dimension, n_hidden=2048, embedding_dim=64, num_embeddings=36,
                                  num_bond_types=num_bond_types, device=device)
        generative_model.load_state_dict(_load_state_dict(edm_weights, device))
        net_dynamics.set_precision(compute_dtype)      # opt-in: "bf16" reduced-precision operands, "f32x6" split-operand fp32
        adj_mat_seer.load_state_dict(_load_state_dict(adj_mat_seer_weights, device))
        self.generative_model = generative_model
        self.set_diffusion_steps(diffusion_steps)
        self.adj_mat_seer = adj_mat_seer
        # (atomic_numbers, coords[n,3]) -> (order, connectivity) | None per molecule: see rdkit_order.py
        self.atom_order_provider = rdkit_order.default_provider() if atom_order_provider == "auto" else atom_order_provider
        # RDKit finish behind the GCN (`redefine_bonds` + `standardize_mol`): "auto" = RDKit's where RDKit imports, else None
        # (the labelled proxy filter); a `host_pool.TaskRef` = the caller's chunk function; None = the proxy filter
        self.finisher = ("rdkit" if HAVE_RDKIT else None) if finisher == "auto" else finisher
        # host fan-out of the two RDKit stages (host_pool.py): worker PROCESSES, created at first use; 0 = this thread
        self.n_host_workers = host_pool.default_workers() if n_host_workers is None else int(n_host_workers)
        if self.cpu_affinity and n_host_workers is None:
            self.n_host_workers = max(1, min(self.n_host_workers, len(self.cpu_affinity)))     # one worker per pinned core at most
        self.task_timeout_s = task_timeout_s      # deadline of one host task (a chunk of molecules); None = none
        self._finish_stage = None    # rdkit_finish.FinishStage of the shard generated last (consumed by the callers)
        self.last_host_order_ms = None    # wall time from the first order task to the last group's hand-off launch
        self.last_host_finish_ms = None   # wall time spent waiting for finish results after the last group was submitted
        self.last_batch = None       # tensors of the most recent generation (x, h, n_nodes, bond)
        self.last_order = None
        self.last_valid_fraction = None   # share of the last batch that passed the validity proxy
        self.last_noise_seed = None       # device-generator seed of the last sharded call on this rank
        self.last_host_assembly_ms = None # host time of the last sharded call's D2H + molecule records
        self._timing = None          # bench.py: {"sampler_start", "sampler_end"} events recorded around the sampler

    def set_diffusion_steps(self, diffusion_steps: int) -> None:
        """Re-wire the noise schedule to `diffusion_steps` denoising steps, exactly as the reference's constructor does
        (conformer_generator.py:105-113); weights, plans and captured graphs are untouched."""
        gm = self.generative_model
        gm.gamma = PredefinedNoiseSchedule(timesteps=diffusion_steps, precision=NOISE_PRECISION)
        gm.time_steps = torch.flip(torch.arange(0, diffusion_steps), dims=[0])
        gm.T = diffusion_steps
        self.diffusion_steps = diffusion_steps

    # ------------------------------------------------------------------ EDM stage
    @torch.no_grad()
    def edm_tensors(self, reference_context: torch.Tensor, n_samples: int = 100, max_n_nodes: int = 32,
                    min_n_nodes: int = 25, resample_steps: int = 0, fixed_fragment=None,
                    inertial_fragment_matching: bool = True, blend_power: int = 3, ifm_diffusion_level: int = 50,
                    sizes: Optional[torch.Tensor] = None):
        """Tensor form of `edm_samples` (conformer_generator.py:125-266): x[B,N,3], h[B,N,8], node_mask.
        `sizes`: molecule sizes drawn by the caller (the sharded path) instead of here."""
        min_n_nodes = max(min_n_nodes, self.min_n_nodes)            # :156-160
        max_n_nodes = min(max_n_nodes, self.max_n_nodes)
        node_mask, edge_mask, batch_context = prepare_edm_input(
            n_samples=n_samples, reference_context=reference_context, context_norms=self.context_norms,
            min_n_nodes=min_n_nodes, max_n_nodes=max_n_nodes, device=self.device, sizes=sizes)
        gm = self.generative_model
        ev = self._timing
        if ev is not None:
            ev["sampler_start"].record()
        if fixed_fragment is None:
            x, h = gm(node_mask, edge_mask, batch_context, resample_steps)
        elif inertial_fragment_matching:
            # generate the complementary fragments separately, then merge (:179-240)
            n_nodes = torch.sum(node_mask, dim=1).to(torch.long)
            ff_x, ff_h = ifm_get_xh_from_fragment(fixed_fragment, self.device)
            f_nm, f_em, f_ctx, shift, rotation = ifm_prepare_gen_fragment_context(
                fixed_fragment_x=ff_x, reference_context=reference_context, n_nodes=n_nodes,
                context_norms=self.context_norms, max_n_nodes=max_n_nodes, min_n_nodes=min_n_nodes,
                device=self.device)
            xg, hg = gm(f_nm, f_em, f_ctx, resample_steps)
            # inverse_coord_transform + ifm_prepare_fragments_for_merge: one HIP launch on the device outputs
            z_known, fixed_mask = ifm_merge_hip(ff_x, ff_h, xg, hg, shift, rotation, self.device, max_n_nodes)
            x, h = gm.merge_fragments(node_mask=node_mask, edge_mask=edge_mask, fixed_mask=fixed_mask,
                                      context=batch_context, z_known=z_known, diffusion_level=ifm_diffusion_level,
                                      resample_steps=resample_steps, blend_power=blend_power)
        else:
            z_known, fixed_mask = prepare_fragment(n_samples=n_samples, fixed_fragment=fixed_fragment,
                                                   max_n_nodes=max_n_nodes, min_n_nodes=min_n_nodes,
                                                   device=self.device)
            x, h = gm.inpaint(node_mask, edge_mask, batch_context, z_known, fixed_mask, resample_steps, blend_power)
        if ev is not None:
            ev["sampler_end"].record()
        return x, h, node_mask

    @torch.no_grad()
    def edm_samples(self, reference_context: torch.Tensor, n_samples: int = 100, max_n_nodes: int = 32,
                    min_n_nodes: int = 25, resample_steps: int = 0, fixed_fragment=None,
                    inertial_fragment_matching: bool = True, blend_power: int = 3, ifm_diffusion_level: int = 50):
        """Samples without bonds (conformer_generator.py:125-266): `List[Chem.Mol]` through `samples_to_rdkit_mol`'s XYZ
        route where RDKit imports (`rdkit_finish.samples`, untested offline), else `GeneratedMolecule` records.  Atoms in
        generation order, as in the reference (canonicalisation happens later, in `generate_conformers`)."""
        x, h, node_mask = self.edm_tensors(reference_context, n_samples, max_n_nodes, min_n_nodes, resample_steps,
                                           fixed_fragment, inertial_fragment_matching, blend_power,
                                           ifm_diffusion_level)
        n_nodes = node_mask.sum(1).reshape(-1).to(torch.long)
        el, _, _ = prepare_adj_mat_seer_input_hip(x, h, n_nodes, self.dimension)
        no_bonds = torch.zeros(x.shape[0], self.dimension, self.dimension, dtype=torch.int8, device=x.device)
        mols = assemble_molecules(x, el, no_bonds, n_nodes)
        for m in mols:
            m.valid = True          # no bonds yet: the connectivity proxy does not apply
        if HAVE_RDKIT:
            from . import rdkit_finish
            return rdkit_finish.samples(mols, self._executor())
        return mols

    # ------------------------------------------------------------------ full pipeline
    def _reference_context(self, reference_conformer, reference_context, n_atoms):
        if reference_conformer is not None and reference_conformer is not False:
            if HAVE_RDKIT and not isinstance(reference_conformer, (str, torch.Tensor)):
                mol = Chem.RemoveHs(reference_conformer)                       # :302-307
                ref_coord = torch.tensor(mol.GetConformer().GetPositions(), dtype=torch.float32)
            elif isinstance(reference_conformer, str):
                ref_coord, _ = parse_molblock_heavy_atoms(reference_conformer)
            else:
                ref_coord = torch.as_tensor(reference_conformer, dtype=torch.float32)
            ref_n_atoms = int(ref_coord.shape[0])
            ref_coord = ref_coord - torch.mean(ref_coord, dim=0)              # :310-311
            ref_context, _ = get_context_shape(ref_coord)
            return ref_context, ref_n_atoms
        if reference_context is not None:
            if not n_atoms:
                raise ValueError(
                    "Reference Number of Atoms should be provided, when generating samples using context.")
            return reference_context, n_atoms
        raise ValueError(
            "Either a reference RDkit Mol object or context as torch.Tensor should be provided for generation.")

    def _executor(self):
        """Where the poolable host tasks run: the process-wide `HostPool` with `n_host_workers` workers and this generator's
        task deadline (workers created at the first submit or by `prestart`), or this thread."""
        return host_pool.shared_pool(self.n_host_workers, getattr(self, "task_timeout_s", host_pool.DEFAULT_TASK_TIMEOUT_S))

    def _generate_shard(self, ref_context, ref_n_atoms: int, variance: int, sizes: Optional[torch.Tensor],
                        n_samples: int, resample_steps, fixed_fragment, inertial_fragment_matching, blend_power,
                        ifm_diffusion_level, optimise_geometry: bool = True):
        """Sampler -> hand-off -> GCN -> bond write-back + validity proxy for `n_samples` molecules on THIS device.
        Returns per-sample DEVICE tensors (dim 0 = n_samples; n_samples may be 0 for an empty shard).

        With an atom-order provider and / or a finisher (RDKit's where RDKit imports) the part behind the sampler is a
        pipeline over groups of molecules (conformer_generator.py:343-366 runs it one molecule at a time): every order
        task of the batch is submitted to the host pool as soon as x, h are on the host; the hand-off + GCN + write-back
        of group g is launched when ITS order results are in; its records are copied back and its finish tasks submitted
        while the pool still works on the order of the later groups.  `self._finish_stage` then holds the pending finish
        (consumed by `generate_conformers*`)."""
        N = min(ref_n_atoms + variance, self.max_n_nodes)
        D = self.dimension
        dev = self.device
        self._finish_stage = None
        self.last_host_order_ms = self.last_host_finish_ms = None
        provider, finisher = self.atom_order_provider, getattr(self, "finisher", None)
        staged = provider is not None or finisher is not None
        executor = self._executor() if staged else None
        if executor is not None:
            # worker start-up (~0.3 s for 32 interpreters, plus `import rdkit` in each) hides under the sampler launched below
            tasks = [rdkit_order.provider_task(provider)[0]] if provider is not None and rdkit_order.provider_task(provider) else []
            if finisher == "rdkit":
                tasks.append(host_pool.FINISH_TASK)
            elif isinstance(finisher, host_pool.TaskRef):
                tasks.append(finisher)
            executor.prestart(tasks)
        if n_samples == 0:
            if finisher is not None:      # an empty shard still takes part in the object gather of the finished molecules
                from . import rdkit_finish
                self._finish_stage = rdkit_finish.FinishStage(finisher, optimise_geometry, executor)
            return dict(x=torch.zeros(0, N, 3, device=dev), elements=torch.zeros(0, D, dtype=torch.int8, device=dev),
                        bond=torch.zeros(0, D, D, dtype=torch.int8, device=dev),
                        n_nodes=torch.zeros(0, dtype=torch.int32, device=dev),
                        valid=torch.zeros(0, dtype=torch.uint8, device=dev))
        x, h, node_mask = self.edm_tensors(
            reference_context=ref_context, n_samples=n_samples, min_n_nodes=ref_n_atoms - variance,
            max_n_nodes=ref_n_atoms + variance, resample_steps=resample_steps, fixed_fragment=fixed_fragment,
            inertial_fragment_matching=inertial_fragment_matching, blend_power=blend_power,
            ifm_diffusion_level=ifm_diffusion_level, sizes=sizes)
        n_nodes = node_mask.sum(1).reshape(-1).to(torch.long)
        groups = rdkit_order.launch_groups(n_samples) if staged else [(0, n_samples)]
        t0 = time.perf_counter()
        # host: RDKit's (or the caller's) order + connectivity - every task of the batch goes out now
        stage = rdkit_order.OrderStage(provider, x, h, n_nodes, executor, groups) if provider is not None else None
        fin = None
        if finisher is not None:
            from . import rdkit_finish
            fin = rdkit_finish.FinishStage(finisher, optimise_geometry, executor)
        parts, orders = [], []
        for g, (lo, hi) in enumerate(groups):
            order = conn = built = None
            if stage is not None:
                order, conn, built = stage.result(g)
            ng = n_nodes[lo:hi]
            # atoms, distances, connectivity AND coordinates come out in the order the GCN sees (canonicalised_samples)
            el, dm, am, x_out = prepare_adj_mat_seer_input_hip(x[lo:hi], h[lo:hi], ng, D, order=order, connectivity=conn,
                                                               with_coords=True)
            bond = self.adj_mat_seer.bond_orders(el, dm, am)
            sym, valid = bond_writeback_hip(bond, el, ng)
            if built is not None and not all(built):          # `MolFromXYZBlock` returned None: the reference drops it
                valid = valid & torch.tensor(built, dtype=torch.bool, device=valid.device)
            parts.append((x_out, el, bond, sym, valid))
            orders.append(order if order is not None else [None] * (hi - lo))
            if fin is not None:                               # this group's records -> the pool, behind the order tasks
                fin.add(molecules_from_tensors(x_out, el.to(torch.int8), sym, ng.to(torch.int32), valid.to(torch.uint8)))
        if stage is not None:
            self.last_host_order_ms = (time.perf_counter() - t0) * 1e3
        cat = (lambda k: parts[0][k]) if len(parts) == 1 else (lambda k: torch.cat([p[k] for p in parts], dim=0))
        x_out, el, bond, sym, valid = (cat(k) for k in range(5))
        self._finish_stage = fin
        self.last_batch = dict(x=x, h=h, n_nodes=n_nodes, elements=el, bond=bond, x_ordered=x_out)     # tensors only
        flat = [o for grp in orders for o in grp]
        self.last_order = None if all(o is None for o in flat) else flat    # per-molecule atom orders (None: generation order)
        return dict(x=x_out, elements=el.to(torch.int8), bond=sym, n_nodes=n_nodes.to(torch.int32),
                    valid=valid.to(torch.uint8))

    def _collect_finish(self, stage):
        """The pending finish of the shard generated last -> per-sample results (None = dropped)."""
        t0 = time.perf_counter()
        done = stage.results()
        self.last_host_finish_ms = (time.perf_counter() - t0) * 1e3
        return done

    @torch.no_grad()
    def generate_conformers(self, reference_conformer=None, n_samples: int = 10, variance: int = 2,
                            reference_context: torch.Tensor = None, n_atoms: int = None,
                            optimise_geometry: bool = True, resample_steps: int = 0, fixed_fragment=None,
                            inertial_fragment_matching: bool = True, blend_power: int = 3,
                            ifm_diffusion_level: int = 50) -> List:
        """Generate molecules from a reference shape (conformer_generator.py:268-368).
        Returns the VALID molecules only (invalid ones are dropped, as in the reference): `List[Chem.Mol]` through the
        reference's standardisation (+ MMFF94 when `optimise_geometry`) where RDKit imports, else `GeneratedMolecule`
        records that pass the labelled valence / single-fragment proxy of `mcg_bond_writeback` (no MMFF: warned once)."""
        ref_context, ref_n_atoms = self._reference_context(reference_conformer, reference_context, n_atoms)
        res = self._generate_shard(ref_context, ref_n_atoms, variance, None, n_samples, resample_steps, fixed_fragment,
                                   inertial_fragment_matching, blend_power, ifm_diffusion_level, optimise_geometry)
        stage, self._finish_stage = getattr(self, "_finish_stage", None), None
        if stage is not None:                     # finished per group while the later groups were still on their way
            done = self._collect_finish(stage)
            kept = [m for m in done if m is not None]
            self.last_valid_fraction = len(kept) / len(done) if done else 0.0
            return kept
        mols = molecules_from_tensors(res["x"], res["elements"], res["bond"], res["n_nodes"], res["valid"])   # single D2H
        kept, self.last_valid_fraction = _finish(mols, optimise_geometry, use_rdkit=False)    # no finisher: the proxy
        return kept

    @torch.no_grad()
    def generate_conformers_sharded(self, reference_conformer=None, n_samples: int = 10, variance: int = 2,
                                    reference_context: torch.Tensor = None, n_atoms: int = None,
                                    optimise_geometry: bool = True, resample_steps: int = 0, fixed_fragment=None,
                                    inertial_fragment_matching: bool = True, blend_power: int = 3,
                                    ifm_diffusion_level: int = 50, group=None, seed: Optional[int] = None,
                                    gather: str = "all", balance: str = "cost") -> List:
        """`generate_conformers` for `n_samples` molecules in TOTAL, sharded over the ranks of the initialised
        `torch.distributed` group (one process per GPU, each with its own generator instance / weight replica;
        SURVEY.md section 8e).  The global size vector is drawn once on rank 0 and broadcast; every rank derives the same
        assignment of molecules to ranks from it (`distributed.assign_shards`: `balance="cost"` = longest-processing-time
        on the edge count n(n-1), "count" = contiguous equal-count slices) and generates its own; ONE all-gather of the
        result tensors at the end gives every rank the full batch, in SAMPLE order.  `seed`: per-rank noise seed `seed + rank` for the
        device generator; None (default) = a base seed drawn on rank 0 and broadcast, so that ranks NEVER share a noise
        stream (every process starts its device generator from the same constant).  A rank whose shard fails makes
        every rank raise `distributed.ShardError`.  `gather="rank0"`: only rank 0 of the group receives (and returns) the
        whole batch - one `gather` instead of the all-gather - and every other rank returns its own shard's molecules.
        Where a finisher runs (RDKit's), every rank finishes ITS OWN shard in its own host pool and the finished Mols are
        gathered as objects.  Without an initialised group this is `generate_conformers`."""
        if gather not in ("all", "rank0"):
            raise ValueError("gather must be 'all' or 'rank0'")
        ref_context, ref_n_atoms = self._reference_context(reference_conformer, reference_context, n_atoms)
        lo_n = max(ref_n_atoms - variance, self.min_n_nodes)
        hi_n = min(ref_n_atoms + variance, self.max_n_nodes)
        self._finish_stage = None
        # decided from the configuration, not from what this rank's shard happened to produce: every rank takes the same
        # path through the collectives (an EMPTY shard has no stage of its own making)
        finishing = getattr(self, "finisher", None) is not None
        finished: List = [None]

        def run_shard(sizes_shard, index):
            res = self._generate_shard(ref_context, ref_n_atoms, variance, sizes_shard, int(index.numel()), resample_steps,
                                       fixed_fragment, inertial_fragment_matching, blend_power, ifm_diffusion_level,
                                       optimise_geometry)
            if finishing:
                # collected HERE, in front of the status exchange: a finish that raises on one rank (a dead worker, a custom
                # finisher's exception, a result-count mismatch) makes EVERY rank raise ShardError instead of leaving the
                # others parked in the object gather
                stage, self._finish_stage = getattr(self, "_finish_stage", None), None
                done = self._collect_finish(stage) if stage is not None else []
                if len(done) != int(index.numel()):
                    raise ValueError(f"the finish returned {len(done)} results for a shard of {int(index.numel())}")
                finished[0] = done
            return res

        def seed_device(s):
            self.last_noise_seed = s          # this rank's noise stream (base seed + rank)
            if self.device.type == "cuda":
                with torch.cuda.device(self.device):
                    torch.cuda.manual_seed(s)

        dst = 0 if gather == "rank0" else None
        _, res, shards = mcg_dist.sharded_generate(
            n_samples, lambda: mcg_dist.draw_global_sizes(n_samples, lo_n, hi_n, group), run_shard, group=group,
            seed=seed, seed_fn=seed_device, gather_dst=dst, balance=balance, gather_tensors=not finishing)
        self.last_shards = shards
        self._finish_stage = None
        if finishing:                         # RDKit's gate: this rank's shard was finished above, the Mols travel as objects
            done = mcg_dist.gather_objects(finished[0], shards, group, dst)
            self.last_host_assembly_ms = 0.0
            kept = [m for m in done if m is not None]
            self.last_valid_fraction = len(kept) / len(done) if done else 0.0
            return kept
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)       # (the D2H copies below would wait for the device anyway: keep that out of the figure)
        t0 = time.perf_counter()
        mols = molecules_from_tensors(res["x"], res["elements"], res["bond"], res["n_nodes"], res["valid"])
        self.last_host_assembly_ms = (time.perf_counter() - t0) * 1e3      # D2H of the gathered tensors + record views
        kept, self.last_valid_fraction = _finish(mols, optimise_geometry, use_rdkit=False)    # no finisher: the proxy
        return kept

    @torch.no_grad()
    def forward(self, reference_conformer=None, n_samples: int = 10, variance: int = 2,
                reference_context: torch.Tensor = None, n_atoms: int = None, optimise_geometry: bool = True,
                resample_steps: int = 0, fixed_fragment=None, inertial_fragment_matching: bool = True,
                blend_power: int = 3, ifm_diffusion_level: int = 50) -> List:
        return self.generate_conformers(reference_conformer, n_samples, variance, reference_context, n_atoms,
                                        optimise_geometry, resample_steps, fixed_fragment,
                                        inertial_fragment_matching, blend_power, ifm_diffusion_level)
