"""Multi-GPU sharding of a generation batch: one process per GPU, no data-path collective.

Every sample is independent through the whole path, so a batch of
`n_samples` is split contiguously over the ranks, each rank runs the full sampler + GCN on
its shard with its own weight replica, and ONE collective at the very end gathers the small
result tensors (x f32, atom types int8, bond orders int8, sizes int32, validity uint8) - RCCL
over xGMI on the GPU box (`backend="nccl"`), gloo in the CPU tests.

Molecule sizes: the GLOBAL size vector is drawn ONCE, on rank 0, from the CPU global RNG with the
reference's own draw (`torch.randint(min, max + 1, (n_samples,))`, mol_utils.py:275) and broadcast
(control plane: n_samples int64).  Every rank then derives the SAME assignment of molecules to ranks from it
(`assign_shards`): longest-processing-time-first on the edge count n(n-1) - the denoiser's cost per molecule - so that a
ragged batch (15..39 atoms: a 7x spread in cost) loads the ranks within ~1 % of each other instead of the +-5 % of equal
COUNTS; the gathered results are put back into sample order.  A seeded single-process run and a sharded run therefore
generate the same molecule sizes in the same order.
Noise: per-rank device generator, seed `seed + rank` (with `seed=None` the base seed is drawn on rank 0 and
broadcast - ranks never share a noise stream); bit-identity of the NOISE with an unsharded run is not promised
(the reference draws it as one [B,N,*] tensor).
Failure: a rank whose shard raises does not leave the others parked in the gather - one status byte per rank is
exchanged first and EVERY rank raises `ShardError`.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_samples: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of rank's shard; the first n_samples % world ranks get one extra."""
    base, extra = divmod(n_samples, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_samples: int, world: int) -> List[int]:
    return [shard_range(n_samples, r, world)[1] - shard_range(n_samples, r, world)[0] for r in range(world)]


def molecule_cost(sizes: torch.Tensor) -> torch.Tensor:
    """Relative cost of a molecule of n atoms on the hot path: its n(n-1) directed edges (27 fused edge-MLP launches per
    denoiser call, ~80 % of the GPU time; SURVEY.md section 8a FLOP accounting)."""
    n = sizes.to(torch.int64)
    return n * (n - 1)


def assign_shards(sizes: torch.Tensor, world: int, balance: str = "cost") -> List[torch.Tensor]:
    """Which samples each rank generates: `world` ascending index vectors that partition range(len(sizes)).
    "cost": longest-processing-time-first on `molecule_cost` (next-heaviest molecule to the least-loaded rank, ties to the
    lower rank - deterministic, identical on every rank; equal sizes degenerate to equal counts, index % world);
    "count": the contiguous equal-count slices of `shard_range`."""
    n = int(sizes.numel())
    if balance == "count":
        return [torch.arange(*shard_range(n, r, world)) for r in range(world)]
    if balance != "cost":
        raise ValueError("balance must be 'cost' or 'count'")
    import heapq
    cost = molecule_cost(sizes.reshape(-1)).tolist()
    order = sorted(range(n), key=lambda i: (-cost[i], i))
    heap = [(0, r) for r in range(world)]
    mine: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        load, r = heapq.heappop(heap)
        mine[r].append(i)
        heapq.heappush(heap, (load + cost[i], r))
    return [torch.tensor(sorted(m), dtype=torch.int64) for m in mine]


def rank_seed(seed: int, rank: int) -> int:
    """Per-rank RNG seed (noise is drawn per shard; bit-identity with an unsharded run is not
    promised - the reference draws noise as one [B,N,*] tensor)."""
    return seed + rank


def world_and_rank(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _solo(world: int) -> bool:
    """A 1-rank world skips every collective - unless MCG_FORCE_COLLECTIVE=1 asks for them (RCCL smoke check on one GPU:
    tools/rccl_one_rank_check.py)."""
    return world == 1 and not os.environ.get("MCG_FORCE_COLLECTIVE")


def _collective_device(group=None) -> torch.device:
    """Where a tensor has to live for this group's collectives: RCCL moves device memory, gloo host memory."""
    if dist.get_backend(group) == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def draw_global_sizes(n_samples: int, min_n_nodes: int, max_n_nodes: int, group=None) -> torch.Tensor:
    """The sizes of the WHOLE batch [n_samples] int64 (CPU), identical on every rank: drawn on rank 0 with
    the reference's draw from the CPU global RNG, then broadcast."""
    world, rank = world_and_rank(group)
    if _solo(world) or not (dist.is_available() and dist.is_initialized()):
        return torch.randint(min_n_nodes, max_n_nodes + 1, (n_samples,))
    dev = _collective_device(group)
    if rank == 0:
        sizes = torch.randint(min_n_nodes, max_n_nodes + 1, (n_samples,)).to(dev)
    else:
        sizes = torch.empty(n_samples, dtype=torch.long, device=dev)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast(sizes, src=src, group=group)
    return sizes.cpu()


def gather_results(local: Dict[str, torch.Tensor], n_samples: int, group=None, dst: Optional[int] = None,
                   shards: Optional[List[torch.Tensor]] = None) -> Dict[str, torch.Tensor]:
    """Gather per-sample result tensors (dim 0 = sample) of unequal shard sizes - the ONE data-path collective.
    `dst` None: all-gather, every rank returns the full batch in rank order.  `dst` = a group rank: `gather` to that rank
    only; it returns the full batch, every other rank its OWN shard (no replicated D2H copy / record assembly on the
    ranks that do not consume the batch).  Shards are padded to the largest shard so that one fixed-size collective per
    tensor suffices.  `shards`: the sample indices of every rank (`assign_shards`; default = contiguous equal-count
    slices); the gathered rows are returned in SAMPLE order."""
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    if _solo(world):
        return local
    rank = dist.get_rank(group)
    sizes = shard_sizes(n_samples, world) if shards is None else [int(ix.numel()) for ix in shards]
    cap = max(sizes)
    perm = None if shards is None else torch.cat(list(shards))      # row k of the rank-ordered concat is sample perm[k]
    # gloo (CPU tests, or a multi-process dry run on one GPU) cannot move device tensors
    via_host = dist.get_backend(group) == "gloo"
    out = {}
    for key, t in local.items():
        src = t.cpu() if via_host else t
        pad = torch.zeros((cap,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        pad[: src.shape[0]] = src
        if dst is None:
            buf = torch.empty((world * cap,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
            dist.all_gather_into_tensor(buf, pad.contiguous(), group=group)
            parts = [buf[r * cap: r * cap + sizes[r]] for r in range(world)]
        else:
            recv = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
            dist.gather(pad.contiguous(), gather_list=recv, dst=dist.get_global_rank(group, dst) if group is not None else dst,
                        group=group)
            if rank != dst:
                out[key] = t
                continue
            parts = [recv[r][: sizes[r]] for r in range(world)]
        full = torch.cat(parts, dim=0)
        if perm is not None:
            full = torch.empty_like(full).index_copy_(0, perm.to(full.device), full)
        out[key] = full.to(t.device) if via_host else full
    return out


class ShardError(RuntimeError):
    """Raised on EVERY rank when the shard of at least one rank failed (a shard failure must
    not leave the other ranks parked in the final collective)."""


def draw_base_seed(group=None) -> int:
    """A noise base seed that is identical on every rank: drawn on rank 0 from the CPU global RNG (after the size
    draw, so the sizes stay what a single-process run draws) and broadcast."""
    world, rank = world_and_rank(group)
    if _solo(world) or not (dist.is_available() and dist.is_initialized()):
        return int(torch.randint(0, 2 ** 31 - 1, (1,)))
    dev = _collective_device(group)
    t = torch.randint(0, 2 ** 31 - 1, (1,)).to(dev) if rank == 0 else torch.zeros(1, dtype=torch.long, device=dev)
    src = dist.get_global_rank(group, 0) if group is not None else 0
    dist.broadcast(t, src=src, group=group)
    return int(t.item())


def exchange_status(ok: bool, group=None) -> List[bool]:
    """One byte per rank, all-gathered BEFORE the result tensors: which ranks finished their shard."""
    world, _ = world_and_rank(group)
    if _solo(world) or not (dist.is_available() and dist.is_initialized()):
        return [ok]
    dev = _collective_device(group)
    mine = torch.tensor([1 if ok else 0], dtype=torch.uint8, device=dev)
    buf = torch.empty(world, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(buf, mine, group=group)
    return [bool(v) for v in buf.cpu().tolist()]


def gather_objects(local: List, shards: List[torch.Tensor], group=None, dst: Optional[int] = None) -> List:
    """Per-sample PYTHON objects (the finished `Chem.Mol`s of a rank's own shard, or None for a dropped molecule) into
    sample order - control plane, pickled by `torch.distributed`; a few hundred bytes per molecule, once per call.
    `dst` None: every rank gets the whole list; else only group rank `dst` does and the others get their own list back."""
    world, rank = world_and_rank(group)
    if _solo(world) or not (dist.is_available() and dist.is_initialized()):
        return local
    if dst is None:
        parts: List = [None] * world
        dist.all_gather_object(parts, local, group=group)
    else:
        parts = [None] * world if rank == dst else None
        dist.gather_object(local, parts, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        if rank != dst:
            return local
    n = sum(int(ix.numel()) for ix in shards)
    out: List = [None] * n
    for r, ix in enumerate(shards):
        if len(parts[r]) != int(ix.numel()):
            raise ValueError(f"rank {r} returned {len(parts[r])} objects for a shard of {int(ix.numel())}")
        for k, i in enumerate(ix.tolist()):
            out[i] = parts[r][k]
    return out


def sharded_generate(n_samples: int, draw_sizes: Callable[[], torch.Tensor],
                     run_shard: Callable[[torch.Tensor, torch.Tensor], Dict[str, torch.Tensor]], group=None,
                     seed: Optional[int] = None, seed_fn: Optional[Callable[[int], None]] = None,
                     gather_dst: Optional[int] = None, balance: str = "cost", gather_tensors: bool = True
                     ) -> Tuple[torch.Tensor, Dict[str, torch.Tensor], List[torch.Tensor]]:
    """The sharded generation step every multi-GPU entry point goes through
    (`MLConformerGenerator.generate_conformers_sharded`, `bench.py --gpus N`):

      sizes  = draw_sizes() once for the whole batch (identical on every rank, see `draw_global_sizes`)
      shards = assign_shards(sizes, world, balance)   # cost-balanced (n(n-1), LPT) by default; same on every rank
      seed_fn(rank_seed(seed, rank))            # per-rank noise stream - ALWAYS when world > 1: with `seed` None the
                                                # base seed is drawn on rank 0 and broadcast (`draw_base_seed`); every
                                                # process starts its device generator from the same constant, so
                                                # "leave the generators alone" would make all shards identical
      local  = run_shard(sizes[mine], mine)     # dict of per-sample tensors, dim 0 = len(mine) (may be 0)
      status = exchange_status(...)             # one byte per rank; ShardError on EVERY rank if any shard failed
      full   = gather_results(local)            # the ONLY data-path collective, at the very end

    Returns (sizes, full, shards) on every rank; `full` is in SAMPLE order whatever the assignment.  `gather_dst` = a group
    rank: only that rank receives the full batch (`gather`), the others get their own shard back (rows in the order of
    `shards[rank]`)."""
    world, rank = world_and_rank(group)
    sizes = draw_sizes()
    if sizes.numel() != n_samples:
        raise ValueError(f"draw_sizes() returned {sizes.numel()} sizes for n_samples={n_samples}")
    shards = assign_shards(sizes, world, balance)
    mine = shards[rank]
    if seed_fn is not None:
        if seed is None and world > 1:
            seed = draw_base_seed(group)
        if seed is not None:
            seed_fn(rank_seed(seed, rank))
    local, failure = None, None
    try:
        local = run_shard(sizes[mine], mine)
        for key, t in local.items():
            if t.shape[0] != mine.numel():
                raise ValueError(f"run_shard returned {t.shape[0]} rows of `{key}` for a shard of {mine.numel()}")
    except Exception as e:  # noqa: BLE001 - reported to every rank below, then re-raised
        if world == 1:
            raise
        failure = e
    status = exchange_status(failure is None, group)
    if not all(status):
        bad = [r for r, ok in enumerate(status) if not ok]
        msg = f"shard generation failed on rank(s) {bad} of {world}"
        if failure is not None:
            raise ShardError(f"{msg}; this rank ({rank}): {type(failure).__name__}: {failure}") from failure
        raise ShardError(msg + f"; this rank ({rank}) finished its shard")
    if not gather_tensors:
        return sizes, local, shards
    return sizes, gather_results(local, n_samples, group, dst=gather_dst, shards=shards), shards
