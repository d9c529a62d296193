"""N > 1 path on CPU: world_size-2 gloo processes shard a batch and gather the results."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ml_conformer_generator_amd.distributed import gather_results, rank_seed, shard_range, shard_sizes


def test_shard_ranges_cover_the_batch():
    for n, w in ((2048, 8), (10, 3), (5, 8), (64, 1)):
        covered = []
        for r in range(w):
            lo, hi = shard_range(n, r, w)
            covered += list(range(lo, hi))
        assert covered == list(range(n))
        assert sum(shard_sizes(n, w)) == n and max(shard_sizes(n, w)) - min(shard_sizes(n, w)) <= 1
    assert rank_seed(7, 3) == 10


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_samples, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_samples, rank, world)
    # each rank "generates" its shard: values encode the global sample index
    idx = torch.arange(lo, hi)
    local = {
        "x": idx.float().view(-1, 1, 1).repeat(1, 5, 3),
        "bond": (idx % 5).to(torch.int8).view(-1, 1, 1).repeat(1, 4, 4),
        "n_nodes": (15 + idx).to(torch.int32),
    }
    full = gather_results(local, n_samples)
    ok = (full["x"].shape[0] == n_samples and torch.equal(full["x"][:, 0, 0], torch.arange(n_samples).float())
          and torch.equal(full["n_nodes"], (15 + torch.arange(n_samples)).to(torch.int32))
          and torch.equal(full["bond"][:, 0, 0], (torch.arange(n_samples) % 5).to(torch.int8)))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_results_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]
