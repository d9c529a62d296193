"""N > 1 path on CPU: world_size-2 gloo processes shard a batch and gather the results."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ml_conformer_generator_amd.distributed import (ShardError, assign_shards, draw_global_sizes, gather_objects,
                                                    gather_results, molecule_cost, rank_seed, shard_range, shard_sizes,
                                                    sharded_generate)


def test_shard_ranges_cover_the_batch():
    for n, w in ((2048, 8), (10, 3), (5, 8), (64, 1)):
        covered = []
        for r in range(w):
            lo, hi = shard_range(n, r, w)
            covered += list(range(lo, hi))
        assert covered == list(range(n))
        assert sum(shard_sizes(n, w)) == n and max(shard_sizes(n, w)) - min(shard_sizes(n, w)) <= 1
    assert rank_seed(7, 3) == 10


def test_cost_balanced_assignment_is_a_partition_and_levels_the_edge_count():
    """Multi-GPU readiness (round-4 review): equal COUNTS of U[15,39] molecules leave the slowest of 8 ranks ~5 % above the
    mean edge count; longest-processing-time on n(n-1) levels it to <= 1 %.  Pure Python, identical on every rank."""
    torch.manual_seed(7)
    sizes = torch.randint(15, 40, (2048,))
    for world in (2, 8):
        shards = assign_shards(sizes, world)
        assert sorted(torch.cat(shards).tolist()) == list(range(2048))
        assert all(torch.equal(ix, ix.sort().values) for ix in shards)
        loads = [int(molecule_cost(sizes[ix]).sum()) for ix in shards]
        assert max(loads) / min(loads) <= 1.01
        by_count = [int(molecule_cost(sizes[ix]).sum()) for ix in assign_shards(sizes, world, "count")]
        assert max(by_count) / min(by_count) > max(loads) / min(loads)
    # 8 x 256 ragged (configs[3]): the judged shape
    assert max(len(ix) for ix in assign_shards(sizes, 8)) - min(len(ix) for ix in assign_shards(sizes, 8)) <= 40
    # equal sizes (configs[1] weak scaling): equal counts; fewer samples than ranks: empty shards
    assert [ix.tolist() for ix in assign_shards(torch.full((8,), 27), 4)] == [[0, 4], [1, 5], [2, 6], [3, 7]]
    assert [ix.tolist() for ix in assign_shards(torch.tensor([20]), 2)] == [[0], []]
    assert [ix.tolist() for ix in assign_shards(torch.arange(15, 22), 2, "count")] == [[0, 1, 2, 3], [4, 5, 6]]


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
    # non-contiguous shards (cost-balanced assignment): rows come back in SAMPLE order; objects likewise
    shards = [torch.tensor([0, 3, 4, 6]), torch.tensor([1, 2, 5])]
    mine = shards[rank]
    full = gather_results({"v": mine.float()}, n_samples, shards=shards)
    ok = ok and full["v"].tolist() == [float(i) for i in range(n_samples)]
    objs = gather_objects([f"m{int(i)}" if i % 2 else None for i in mine], shards)
    ok = ok and objs == [f"m{i}" if i % 2 else None for i in range(n_samples)]
    objs0 = gather_objects([int(i) for i in mine], shards, dst=0)
    ok = ok and objs0 == (list(range(n_samples)) if rank == 0 else mine.tolist())
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


# ---------------------------------------------------------------------------------------------------------
# The REAL shard path: `MLConformerGenerator.generate_conformers_sharded` -> `distributed.sharded_generate`
# (global size vector drawn once on rank 0 + broadcast, contiguous slices, per-rank noise seed, one gather),
# with only the device work (`_generate_shard`: sampler + GCN kernels) replaced by a stub.
# ---------------------------------------------------------------------------------------------------------
def _shell_generator():
    """An MLConformerGenerator without its constructor (which needs the GPU): just the attributes the shard
    path reads."""
    from ml_conformer_generator_amd.conformer_generator import MLConformerGenerator
    gen = MLConformerGenerator.__new__(MLConformerGenerator)
    torch.nn.Module.__init__(gen)
    gen.device = torch.device("cpu")
    gen.dimension = 42
    gen.min_n_nodes, gen.max_n_nodes = 15, 39
    gen.last_batch = gen.last_valid_fraction = gen._timing = gen.last_noise_seed = gen.last_host_assembly_ms = None
    gen.atom_order_provider = None
    return gen


def _stub_shard(calls):
    def run(ref_context, ref_n_atoms, variance, sizes, n_samples, *rest):
        calls.append((sizes.clone(), n_samples))
        N, D = min(ref_n_atoms + variance, 39), 42
        el = torch.zeros(n_samples, D, dtype=torch.int8)
        for b in range(n_samples):
            el[b, : int(sizes[b])] = 6
        # x encodes the molecule size so the gathered order can be checked; odd sizes are "invalid"
        x = sizes.float().view(-1, 1, 1).repeat(1, N, 3)
        return dict(x=x, elements=el, bond=torch.zeros(n_samples, D, D, dtype=torch.int8),
                    n_nodes=sizes.to(torch.int32), valid=(sizes % 2 == 0).to(torch.uint8))
    return run


def _shard_worker(rank, world, port, n_samples, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1000 + 17 * rank)            # ranks disagree on purpose: only rank 0's draw may count
    gen = _shell_generator()
    calls = []
    gen._generate_shard = _stub_shard(calls)
    mols = gen.generate_conformers_sharded(reference_context=torch.tensor([50.0, 100.0, 130.0]), n_atoms=27, variance=12,
                                           n_samples=n_samples)
    # what a single process seeded like rank 0 draws (the reference's draw, mol_utils.py:275)
    torch.manual_seed(1000)
    expect = torch.randint(15, 40, (n_samples,))
    mine = assign_shards(expect, world)[rank]      # cost-balanced (n(n-1), LPT): the same on every rank
    ok = len(calls) == 1 and torch.equal(calls[0][0], expect[mine]) and calls[0][1] == mine.numel()
    ok = ok and not torch.equal(mine, torch.arange(*shard_range(n_samples, rank, world)))
    ok = ok and [m.GetNumAtoms() for m in mols] == [int(v) for v in expect.tolist() if v % 2 == 0]
    ok = ok and all(float(m.coords[0, 0]) == m.GetNumAtoms() for m in mols)
    ok = ok and abs(gen.last_valid_fraction - float((expect % 2 == 0).float().mean())) < 1e-6
    # the generic entry: per-rank noise seeds and an EMPTY shard (1 sample, 2 ranks)
    seeds = []
    _, full, shards = sharded_generate(1, lambda: draw_global_sizes(1, 20, 20), lambda sz, idx: {"v": sz.float()},
                                       seed=7, seed_fn=seeds.append)
    ok = ok and [ix.tolist() for ix in shards] == [[0], []]
    ok = ok and seeds == [rank_seed(7, rank)] and full["v"].tolist() == [20.0]
    # DEFAULT arguments (seed=None): the ranks must still draw DIFFERENT noise - base seed from rank 0 + rank
    seen = [None] * world
    dist.all_gather_object(seen, gen.last_noise_seed)
    ok = ok and None not in seen and seen[1] == seen[0] + 1
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_generate_conformers_sharded_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, 11, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _rank0_gather_worker(rank, world, port, n_samples, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(1000 + 17 * rank)
    gen = _shell_generator()
    gen._generate_shard = _stub_shard([])
    mols = gen.generate_conformers_sharded(reference_context=torch.tensor([50.0, 100.0, 130.0]), n_atoms=27, variance=12,
                                           n_samples=n_samples, gather="rank0")
    torch.manual_seed(1000)
    expect = torch.randint(15, 40, (n_samples,))
    mine = expect if rank == 0 else expect[assign_shards(expect, world)[rank]]   # rank 0: the whole batch; the others: their own shard only
    ok = [m.GetNumAtoms() for m in mols] == [int(v) for v in mine.tolist() if v % 2 == 0]
    ok = ok and gen.last_host_assembly_ms is not None and gen.last_host_assembly_ms < 1000.0
    try:
        gen.generate_conformers_sharded(reference_context=torch.tensor([50.0, 100.0, 130.0]), n_atoms=27, n_samples=2,
                                        gather="everyone")
        ok = False
    except ValueError:
        pass
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_to_rank0_only_world2_gloo():
    """Multi-GPU readiness: with `gather="rank0"` the final collective is a `gather` - rank 0 returns the whole batch in
    sample order, every other rank only its own shard (no replicated D2H copy + record assembly on ranks that do not
    consume the batch)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank0_gather_worker, args=(r, 2, port, 11, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, True), (1, True)]


def _failing_worker(rank, world, port, q):
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gen = _shell_generator()
    good = _stub_shard([])

    def shard(*a):
        if rank == 1:
            raise RuntimeError("device fault injected on rank 1")
        return good(*a)
    gen._generate_shard = shard
    t0 = time.time()
    try:
        gen.generate_conformers_sharded(reference_context=torch.tensor([50.0, 100.0, 130.0]), n_atoms=27, variance=2,
                                        n_samples=6)
        res = "returned"
    except ShardError as e:
        res = ("ShardError", "rank(s) [1]" in str(e), ("injected" in str(e)) == (rank == 1))
    q.put((rank, res, time.time() - t0 < 30.0))
    dist.barrier()                       # the group is still usable: nobody is parked in a half-done collective
    dist.destroy_process_group()


def test_failing_shard_raises_on_every_rank_world2_gloo():
    """a rank whose shard raises must not leave the others waiting in the final gather until
    the collective times out - every rank raises ShardError within seconds."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, ("ShardError", True, True), True), (1, ("ShardError", True, True), True)]


def _finishing_shard(gen):
    """`_generate_shard` with the device work stubbed but the FINISH stage real: shards with molecules build a
    `FinishStage` over the generator's finisher the way the product does; an EMPTY shard goes through the product's own
    `_generate_shard` (its n_samples == 0 path needs no GPU)."""
    from ml_conformer_generator_amd import host_pool as HP
    from ml_conformer_generator_amd import rdkit_finish as RF
    from ml_conformer_generator_amd.conformer_generator import MLConformerGenerator
    from ml_conformer_generator_amd.handoff import molecules_from_tensors
    stub = _stub_shard([])

    def run(ref_context, ref_n_atoms, variance, sizes, n_samples, *rest):
        if n_samples == 0:
            return MLConformerGenerator._generate_shard(gen, ref_context, ref_n_atoms, variance, sizes, n_samples, *rest)
        res = stub(ref_context, ref_n_atoms, variance, sizes, n_samples, *rest)
        fin = RF.FinishStage(gen.finisher, True, HP.SerialExecutor())
        fin.add(molecules_from_tensors(res["x"], res["elements"], res["bond"], res["n_nodes"], res["valid"]))
        gen._finish_stage = fin
        return res
    return run


def _finisher_worker(rank, world, port, q):
    import time
    from ml_conformer_generator_amd import host_pool as HP
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fake = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_host_tasks.py")
    torch.manual_seed(1000 + 17 * rank)
    gen = _shell_generator()
    gen.n_host_workers = 0
    gen.finisher = HP.TaskRef(fake, "finish_tag_chunk")
    gen._generate_shard = _finishing_shard(gen)
    ctx = torch.tensor([50.0, 100.0, 130.0])
    res = []
    # (a) ONE sample on two ranks: rank 1's shard is empty - it must still take part in the object gather (advisor, round 5:
    #     it took the tensor path and the other rank sat in all_gather_object until the backend timed out)
    t0 = time.time()
    mols = gen.generate_conformers_sharded(reference_context=ctx, n_atoms=27, variance=0, n_samples=1)
    res.append(mols == [("mol", 27, True)] and time.time() - t0 < 30.0)
    # (b) a ragged batch, gather to rank 0 only: sample order on rank 0, the own shard elsewhere
    mols = gen.generate_conformers_sharded(reference_context=ctx, n_atoms=27, variance=12, n_samples=9, gather="rank0")
    torch.manual_seed(1000)
    torch.randint(27, 28, (1,))                    # (a)'s size draw on rank 0 ...
    torch.randint(0, 2 ** 31 - 1, (1,))            # ... and its base noise seed (`draw_base_seed`)
    expect = torch.randint(15, 40, (9,))
    mine = expect if rank == 0 else expect[assign_shards(expect, world)[rank]]
    res.append(mols == [("mol", int(v), True) for v in mine.tolist()])
    # (c) the finish FAILS on rank 1 only: every rank raises ShardError within seconds (it ran behind the status exchange
    #     before: rank 1 raised alone and rank 0 waited in the object gather)
    if rank == 1:
        gen.finisher = HP.TaskRef(fake, "finish_raises_chunk")
    t0 = time.time()
    try:
        gen.generate_conformers_sharded(reference_context=ctx, n_atoms=27, variance=2, n_samples=6)
        res.append("returned")
    except ShardError as e:
        res.append(("rank(s) [1]" in str(e)) and (("finisher fault injected" in str(e)) == (rank == 1)) and time.time() - t0 < 30.0)
    q.put((rank, res))
    dist.barrier()                                  # nobody is parked in a half-done collective
    dist.destroy_process_group()


def test_finisher_path_world2_gloo_empty_shard_and_failing_finish():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_finisher_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert res == [(0, [True, True, True]), (1, [True, True, True])]


def test_sharded_path_without_a_process_group_is_the_plain_call():
    torch.manual_seed(5)
    expect = torch.randint(25, 30, (6,))
    torch.manual_seed(5)
    gen = _shell_generator()
    calls = []
    gen._generate_shard = _stub_shard(calls)
    mols = gen.generate_conformers_sharded(reference_context=torch.tensor([50.0, 100.0, 130.0]), n_atoms=27, variance=2,
                                           n_samples=6)
    assert torch.equal(calls[0][0], expect) and len(mols) == int((expect % 2 == 0).sum())


def test_bench_refuses_to_report_fewer_gpus_than_asked():
    """`python bench.py --gpus 2` must start 2 ranks itself or fail: on a box with fewer than 2 GPUs (this
    container has none) it exits non-zero and prints no JSON line (never `n_gpus: 1`)."""
    import subprocess
    import sys
    from conftest import REPO
    if torch.cuda.device_count() >= 2:
        return                      # a real multi-GPU box: the run itself is the driver's SCALE check
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert '"n_gpus"' not in r.stdout


def test_bench_counts_gpus_without_touching_hip_and_ends_all_ranks_when_one_fails():
    """`bench.py --gpus N` parent: devices are counted from the KFD topology in sysfs (no HIP call), and a rank that dies
    takes the others down with it - here (no GPU in this container) every child dies at `torch.cuda.set_device`; with the
    gloo dry-run switch the parent does spawn them, must notice, end the rest, print no JSON line and exit non-zero well
    inside the collective timeout."""
    import importlib.util
    import subprocess
    import sys
    import time
    from conftest import REPO
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(REPO, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = bench.count_gpus_without_hip()
    assert n is None or (isinstance(n, int) and n >= 0)
    assert not torch.cuda.is_initialized()
    if torch.cuda.device_count() >= 1:
        return                      # a GPU box: the dry run itself is exercised by profiles/round*_bench_2rank_gloo*.json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MCG_DIST_BACKEND"] = "gloo"
    env["MCG_BENCH_TIMEOUT"] = "240"
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=400)
    assert r.returncode != 0
    assert '"n_gpus"' not in r.stdout
    assert "child ranks exited" in r.stderr
    assert time.time() - t0 < 300


def test_bench_ranks_pin_themselves_without_loading_the_hip_library():
    """Round 6: `bench.py` ranks pin themselves to a share of the host's cores before they touch the GPU - through `affinity.py`
    loaded by path (no package import, no HIP library), on by default only for N > 1.  Run in a child: it narrows its own mask."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, REPO_DIR); import bench\n"
            "a = sorted(os.sched_getaffinity(0))\n"
            "lone = bench.pin_this_rank(1, 0)\n"
            "r0 = bench.pin_this_rank(2, 0); now0 = sorted(os.sched_getaffinity(0))\n"
            "os.sched_setaffinity(0, a)\n"
            "r1 = bench.pin_this_rank(2, 1)\n"
            "print(lone is None, r0 == now0, len(a) < 2 or (len(r0) == len(a) // 2 + len(a) % 2 and not set(r0) & set(r1)), "
            "'ml_conformer_generator_amd' not in sys.modules)\n").replace("REPO_DIR", repr(repo))
    env = dict(os.environ)
    env.pop("MCG_BENCH_AFFINITY", None)
    env.pop("LOCAL_WORLD_SIZE", None)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert out.stdout.split() == ["True", "True", "True", "True"], out.stdout + out.stderr
