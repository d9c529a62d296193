"""Multi-GPU sharding of a generation batch: one process per GPU, no data-path collective.

Every sample is independent through the whole path, so a batch of
`n_samples` is split contiguously over the ranks, each rank runs the full sampler + GCN on
its shard with its own weight replica, and ONE collective at the very end gathers the small
result tensors (x f32, atom types int8, bond orders int8, sizes int32) - RCCL over xGMI on
the GPU box (`backend="nccl"`), gloo in the CPU tests.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


def shard_range(n_samples: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of rank's shard; the first n_samples % world ranks get one extra."""
    base, extra = divmod(n_samples, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_sizes(n_samples: int, world: int) -> List[int]:
    return [shard_range(n_samples, r, world)[1] - shard_range(n_samples, r, world)[0] for r in range(world)]


def rank_seed(seed: int, rank: int) -> int:
    """Per-rank RNG seed (noise is drawn per shard; bit-identity with an unsharded run is not
    promised - the reference draws noise as one [B,N,*] tensor)."""
    return seed + rank


def gather_results(local: Dict[str, torch.Tensor], n_samples: int, group=None) -> Dict[str, torch.Tensor]:
    """All-gather per-sample result tensors (dim 0 = sample) of unequal shard sizes.
    Every rank returns the full batch in rank order.  Shards are padded to the largest shard
    so that a single fixed-size all_gather per tensor suffices."""
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    if world == 1 and not os.environ.get("MCG_FORCE_COLLECTIVE"):
        return local          # (MCG_FORCE_COLLECTIVE=1: run the collective on a 1-rank group - RCCL smoke check)
    sizes = shard_sizes(n_samples, world)
    cap = max(sizes)
    # gloo (CPU tests, or a multi-process dry run on one GPU) cannot move device tensors
    via_host = dist.get_backend(group) == "gloo"
    out = {}
    for key, t in local.items():
        src = t.cpu() if via_host else t
        pad = torch.zeros((cap,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        pad[: src.shape[0]] = src
        buf = torch.empty((world * cap,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
        dist.all_gather_into_tensor(buf, pad.contiguous(), group=group)
        parts = [buf[r * cap: r * cap + sizes[r]] for r in range(world)]
        full = torch.cat(parts, dim=0)
        out[key] = full.to(t.device) if via_host else full
    return out
