"""Host fan-out of the two RDKit stages (`ml_conformer_generator_amd/host_pool.py`), driven with
fake chunk functions (tests/fake_host_tasks.py) - RDKit exists neither here nor on the GPU boxes; the real RDKit tasks run
in tests/test_rdkit_optional.py where RDKit imports."""
import os
import time

import numpy as np
import pytest
import torch

from ml_conformer_generator_amd import host_pool as HP

FAKE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_host_tasks.py")


def ref(name):
    return HP.TaskRef(FAKE, name)


@pytest.fixture(scope="module")
def pool():
    p = HP.HostPool(8).start()
    yield p
    p.close()


def test_results_keep_item_order_and_none_and_equal_the_serial_path(pool):
    items = list(range(203))
    serial = HP.map_ordered(HP.SerialExecutor(), ref("square_chunk"), items)
    for chunk in (1, 5, 64, None):
        assert HP.map_ordered(pool, ref("square_chunk"), items, chunk=chunk) == serial
    assert serial[3] is None and serial[10] is None and serial[4] == 16          # None = "dropped" survives the trip
    assert HP.map_ordered(pool, ref("square_chunk"), []) == []


def test_workers_are_fresh_processes_without_torch_or_the_package(pool):
    """Never a fork of this process (which may own a GPU context), never the package's __init__: a worker is
    `python _host_worker.py` holding the task file's imports only, one thread each."""
    info = HP.map_ordered(pool, ref("introspect_chunk"), list(range(32)), chunk=1)
    pids = {i["pid"] for i in info}
    assert pids <= set(pool.worker_pids()) and len(pids) >= 2 and os.getpid() not in pids
    for i in info:
        assert i["ppid"] == os.getpid() and i["argv0"].endswith("_host_worker.py")
        assert i["heavy"] == [] and i["omp"] == "1"


def test_eight_workers_are_at_least_six_times_faster_on_a_2ms_per_molecule_stage(pool):
    items = list(range(256))
    HP.map_ordered(pool, ref("sleep_chunk"), items[:16], (0.0,))          # warm: workers started, task file loaded
    t0 = time.perf_counter()
    serial = HP.map_ordered(HP.SerialExecutor(), ref("sleep_chunk"), items, (0.002,))
    t_serial = time.perf_counter() - t0
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        pooled = HP.map_ordered(pool, ref("sleep_chunk"), items, (0.002,))
        best = min(best, time.perf_counter() - t0)
    assert [i for i, _ in pooled] == items == [i for i, _ in serial]
    assert len({pid for _, pid in pooled}) == 8
    assert t_serial / best >= 6.0, f"serial {t_serial * 1e3:.0f} ms, 8 workers {best * 1e3:.0f} ms"


def test_a_task_exception_reaches_the_caller_with_its_type_and_the_pool_survives(pool):
    futs = [pool.submit(ref("raise_on_13"), list(range(lo, lo + 4))) for lo in range(0, 24, 4)]
    got = []
    for f in futs:
        try:
            got.append(f.result(timeout=30))
        except ValueError as e:
            assert "thirteen" in str(e) and "raise_on_13" in e.worker_traceback
            got.append("ValueError")
    assert got == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], "ValueError", [16, 17, 18, 19], [20, 21, 22, 23]]
    with pytest.raises(ValueError):
        HP.map_ordered(pool, ref("raise_on_13"), list(range(20)), chunk=3)
    with pytest.raises(ValueError):                                       # the serial path raises the same way
        HP.map_ordered(HP.SerialExecutor(), ref("raise_on_13"), list(range(20)))
    assert HP.map_ordered(pool, ref("square_chunk"), [2, 3]) == [4, None]


def test_a_dying_worker_fails_its_task_and_is_replaced():
    with HP.HostPool(2) as p:
        before = set(p.worker_pids())
        futs = [p.submit(ref("crash_chunk"), [k]) for k in range(10)]
        out = []
        for f in futs:
            try:
                out.append(f.result(timeout=60))
            except HP.HostPoolError as e:
                assert "died" in str(e)
                out.append("dead")
        assert out == [[0], [1], [2], [3], [4], "dead", [6], [7], [8], [9]]
        assert p.workers_replaced == 1 and len(p.worker_pids()) == 2 and set(p.worker_pids()) != before
        pids = p.worker_pids()
    time.sleep(0.2)
    for pid in pids:                                  # closed pool: its workers are gone (EOF on the task pipe)
        with pytest.raises(OSError):
            os.kill(pid, 0)
    with pytest.raises(HP.HostPoolError):
        p.submit(ref("square_chunk"), [1])


def test_shared_pool_and_chunking_rules():
    assert isinstance(HP.shared_pool(0), HP.SerialExecutor)
    a, b = HP.shared_pool(3), HP.shared_pool(3)
    assert a is b and a.n_workers == 3 and a.worker_pids() == []           # lazily started: nothing spawned yet
    saved = os.environ.pop("LOCAL_WORLD_SIZE", None)
    try:
        assert HP.default_workers() == min(32, os.cpu_count())
        os.environ["LOCAL_WORLD_SIZE"] = "8"              # one process per GPU: the node's cores are shared between its ranks
        assert HP.default_workers() == max(1, min(32, os.cpu_count() // 8))
        os.environ["LOCAL_WORLD_SIZE"] = "not a number"
        assert HP.default_workers() == min(32, os.cpu_count())
    finally:
        os.environ.pop("LOCAL_WORLD_SIZE", None)
        if saved is not None:
            os.environ["LOCAL_WORLD_SIZE"] = saved
    assert HP.chunk_bounds(10, 4) == [(0, 4), (4, 8), (8, 10)] and HP.chunk_bounds(0, 4) == []
    assert HP.task_chunk(64, 32) == 1 and HP.task_chunk(2048, 32) == 8 and HP.task_chunk(256, 8) == 8
    assert HP.task_chunk(5, 0) == 5 and HP.task_chunk(0, 4) == 1
    with pytest.raises(ValueError):
        HP.HostPool(0)
    # a process already restricted to a subset of the cores (pinned rank, cpuset) takes THAT as its share - no second division
    if hasattr(os, "sched_setaffinity") and len(os.sched_getaffinity(0)) >= 2 and len(os.sched_getaffinity(0)) == os.cpu_count():
        import subprocess
        import sys
        code = ("import os, importlib.util\n"
                "spec = importlib.util.spec_from_file_location('hp', %r); hp = importlib.util.module_from_spec(spec); spec.loader.exec_module(hp)\n"
                "os.environ['LOCAL_WORLD_SIZE'] = '8'\n"
                "a = sorted(os.sched_getaffinity(0)); os.sched_setaffinity(0, a[:2])\n"
                "print(hp.default_workers())\n") % HP.__file__
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
        assert out.stdout.strip() == "2", out.stdout + out.stderr


# ------------------------------------------------------------------------------------------------ the two stages
def _batch(B=37, N=24, seed=3):
    g = torch.Generator().manual_seed(seed)
    n = torch.randint(9, N + 1, (B,), generator=g)
    x = torch.randn(B, N, 3, generator=g) * 1.7
    cls = torch.randint(0, 7, (B, N), generator=g)                  # classes 0..6: never Br ...
    cls[5, 0] = 7                                                   # ... except molecule 5: "cannot be built"
    h = torch.nn.functional.one_hot(cls, 8).float()
    return x, h, n


def test_order_stage_pooled_equals_serial_per_group(pool):
    from ml_conformer_generator_amd import rdkit_order as RO
    x, h, n = _batch()
    groups = RO.launch_groups(x.shape[0])
    assert groups[0][0] == 0 and groups[-1][1] == 37 and len(groups) == 4
    assert RO.launch_groups(5) == [(0, 5)] and RO.launch_groups(16) == [(0, 8), (8, 16)]
    serial = RO.OrderStage(ref("order_chunk_no_sleep"), x, h, n, HP.SerialExecutor(), groups)
    pooled = RO.OrderStage(ref("order_chunk"), x, h, n, pool, groups)
    whole = RO.batch_order_and_connectivity(ref("order_chunk_no_sleep"), x, h, n, pool)
    cat_o, cat_c, cat_b = [], [], []
    for g in range(len(groups)):
        (o1, c1, b1), (o2, c2, b2) = serial.result(g), pooled.result(g)
        assert o1 == o2 and b1 == b2 and all(np.array_equal(a, b) for a, b in zip(c1, c2))
        cat_o += o1; cat_c += c1; cat_b += b1
    assert cat_o == whole[0] and cat_b == whole[2] and all(np.array_equal(a, b) for a, b in zip(cat_c, whole[1]))
    assert cat_b[5] is False and cat_o[5] is None and cat_c[5].shape == (int(n[5]), int(n[5])) and sum(cat_b) == 36
    # a plain callable is not poolable: it runs here, one molecule at a time, same results
    import sys
    sys.path.insert(0, os.path.dirname(FAKE))
    import fake_host_tasks as F
    calls = []

    def plain(z, c):
        calls.append(len(z))
        return F.order_chunk_no_sleep([(z, c)])[0]
    o3, c3, b3 = RO.batch_order_and_connectivity(plain, x, h, n, pool)
    assert o3 == whole[0] and b3 == whole[2] and calls == [int(v) for v in n]
    assert RO.provider_task(plain) is None and RO.provider_task(RO.rdkit_provider)[0] == HP.ORDER_TASK


def test_order_stage_errors_bondless_valueerror_and_mixed_connectivity(pool):
    from ml_conformer_generator_amd import rdkit_order as RO
    x, h, n = _batch(B=16)
    x[9] = x[9] * 50.0                                               # nothing within 1.9 A: "no connectivity perceived"
    for ex in (pool, HP.SerialExecutor()):
        st = RO.OrderStage(ref("order_chunk_bondless_raises"), x, h, n, ex, RO.launch_groups(16))
        st.result(0)
        with pytest.raises(ValueError, match="Bonds must be specified"):
            st.result(1)
        with pytest.raises(ValueError, match="connectivity for every molecule or for none"):
            RO.batch_order_and_connectivity(ref("order_chunk_half_connectivity"), x, h, n, ex)


def _records(B=21, seed=4):
    from ml_conformer_generator_amd.handoff import molecules_from_tensors
    g = torch.Generator().manual_seed(seed)
    n = torch.randint(3, 12, (B,), generator=g).to(torch.int32)
    x = torch.randn(B, 12, 3, generator=g)
    el = torch.randint(6, 10, (B, 42), generator=g).to(torch.int8)
    bd = torch.randint(0, 3, (B, 42, 42), generator=g).to(torch.int8)
    bd[4] = 0                                                        # no bonds: the fake gate drops it
    return molecules_from_tensors(x, el, bd, n, torch.ones(B, dtype=torch.uint8))


def test_finish_stage_pooled_equals_serial_and_keeps_sample_order(pool):
    from ml_conformer_generator_amd import rdkit_finish as RF
    recs = _records()
    out = {}
    for name, ex in (("serial", HP.SerialExecutor()), ("pool", pool)):
        st = RF.FinishStage(ref("finish_chunk"), True, ex)
        st.add(recs[:8]); st.add(recs[8:9]); st.add([]); st.add(recs[9:])
        out[name] = st.results()
    assert out["serial"] == out["pool"] and len(out["pool"]) == 21
    assert out["pool"][4] is None and sum(r is None for r in out["pool"]) == 1
    for rec, r in zip(recs, out["pool"]):
        if r is not None:
            assert r["z"] == tuple(rec.atomic_numbers) and r["mmff"] is True
            assert r["xyz"] == "%.9f" % float(np.asarray(rec.coords.tolist(), dtype=np.float64).sum())
    st = RF.FinishStage(ref("finish_chunk"), False, pool)
    st.add(recs[:2])
    assert [r["mmff"] for r in st.results()] == [False, False]
    with pytest.raises(ValueError):
        RF.FinishStage("no such finisher", True, pool)


def test_pipeline_overlaps_order_and_finish_of_different_groups(pool):
    """The generator's loop shape (`_generate_shard`): all order tasks out at once, per group wait -> 'launch' -> finish
    tasks in behind; with 8 workers and 2 ms per molecule and stage, 128 molecules take ~ 2 x 128 x 2 ms / 8 = 64 ms, not
    the 512 ms of the serial loop."""
    from ml_conformer_generator_amd import rdkit_finish as RF
    from ml_conformer_generator_amd import rdkit_order as RO
    x, h, n = _batch(B=128, N=12, seed=8)
    recs = _records(B=128, seed=9)
    groups = RO.launch_groups(128)

    def run(ex, order_name):
        t0 = time.perf_counter()
        st = RO.OrderStage(ref(order_name), x, h, n, ex, groups)
        fin = RF.FinishStage(ref("finish_chunk"), True, ex)
        orders = []
        for g, (lo, hi) in enumerate(groups):
            orders += st.result(g)[0]
            fin.add(recs[lo:hi])
        res = fin.results()
        return orders, res, time.perf_counter() - t0
    run(pool, "order_chunk")                                              # warm
    o_s, r_s, t_s = run(HP.SerialExecutor(), "order_chunk")
    o_p, r_p, t_p = min((run(pool, "order_chunk") for _ in range(3)), key=lambda t: t[2])
    assert o_s == o_p and r_s == r_p
    assert t_s / t_p >= 5.0, f"serial {t_s * 1e3:.0f} ms, pooled pipeline {t_p * 1e3:.0f} ms"


# ------------------------------------------------------------------------------------------------ hardening (round 6)
def test_a_stuck_worker_is_killed_replaced_and_its_molecules_dropped_with_a_warning():
    """Round-5 review: dead workers were replaced, stuck ones were not - a worker that never returns from RDKit parked
    `generate_conformers` forever.  With a deadline the serving thread kills THAT child by handle, replaces it, and the
    chunk's molecules come back as None (= dropped, the reference's "any failure => invalid") - no exception."""
    import warnings
    with HP.HostPool(2, task_timeout_s=0.5) as p:
        before = set(p.worker_pids())
        t0 = time.perf_counter()
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter("always")
            futs = [p.submit(ref("hang_on_7"), [lo, lo + 1]) for lo in range(0, 12, 2)]
            out = [f.result(timeout=30) for f in futs]
        took = time.perf_counter() - t0
        assert out == [[100, 101], [102, 103], [104, 105], [None, None], [108, 109], [110, 111]]
        assert took < 10.0, f"{took:.1f} s: the deadline was not enforced"
        msgs = [str(w.message) for w in seen if issubclass(w.category, RuntimeWarning)]
        assert len(msgs) == 1 and "hang_on_7" in msgs[0] and "0.5 s" in msgs[0] and "dropped" in msgs[0]
        assert p.tasks_timed_out == 1 and p.workers_replaced == 1
        after = set(p.worker_pids())
        assert len(after) == 2 and after != before
        (gone,) = before - after
        time.sleep(0.1)
        with pytest.raises(OSError):                  # the stuck child itself was ended (reaped: no zombie either)
            os.kill(gone, 0)
        assert p.submit(ref("hang_on_7"), [1, 2]).result(timeout=30) == [101, 102]       # the replacement works
        # per-task override: no deadline at all for this one, a long one for that
        assert p.submit(ref("sleep_chunk"), [3], (0.8,), timeout=None).result(timeout=30)[0][0] == 3
        assert p.submit(ref("sleep_chunk"), [4], (0.8,), timeout=5.0).result(timeout=30)[0][0] == 4
        assert p.tasks_timed_out == 1
    with pytest.raises(ValueError):
        HP.HostPool(2, task_timeout_s=0.0)
    # the shared pool is keyed by (workers, deadline); the serial executor takes the argument and ignores it
    assert HP.shared_pool(3, 5.0) is HP.shared_pool(3, 5.0) and HP.shared_pool(3, 5.0) is not HP.shared_pool(3, None)
    assert HP.shared_pool(3).task_timeout_s == HP.DEFAULT_TASK_TIMEOUT_S == 60.0
    assert HP.SerialExecutor().submit(ref("square_chunk"), [2], (), timeout=0.001).result() == [4]


def test_an_exception_class_only_the_worker_can_import_still_resolves_the_future(pool):
    """Advisor (round 5): a reply that does not unpickle in the parent killed the serving thread and left `.result()` waiting
    forever.  The worker now sends such exceptions as RuntimeError naming the type; the pool survives."""
    f = pool.submit(ref("raise_task_local"), [1, 2])
    with pytest.raises(RuntimeError, match="TaskLocalError: only the worker knows this class") as ei:
        f.result(timeout=30)
    assert "raise_task_local" in ei.value.worker_traceback
    assert HP.map_ordered(pool, ref("square_chunk"), [2, 3]) == [4, None]
    assert all(t.is_alive() for t in pool._threads)


def test_workers_do_not_have_the_package_directory_on_sys_path(pool):
    """Advisor (round 5): run by file path, a worker had the package's own directory as sys.path[0] - `config.py`,
    `distributed.py`, `schedule.py` ... shadowed top-level modules of the same name for everything a task imports."""
    pkg = os.path.dirname(os.path.abspath(HP.__file__))
    for path in pool.submit(ref("sys_path_chunk"), [0, 1]).result(timeout=30):
        assert all(os.path.abspath(p or ".") != pkg for p in path), path


def test_prestart_spawns_the_workers_in_the_background_and_submit_waits_for_it():
    """Round-5 review: the pool started at the first submit - AFTER the sampler - so the first call paid the workers'
    start-up serially.  `prestart()` returns at once; the generator calls it before it launches the sampler."""
    p = HP.HostPool(4)
    try:
        t0 = time.perf_counter()
        p.prestart([ref("square_chunk")])              # + the task file: the workers import it (numpy) as they come up
        p.prestart()                                   # idempotent while the start is in progress
        assert time.perf_counter() - t0 < 0.5 and HP.SerialExecutor().prestart() is None       # (returns at once: spawning 4 interpreters takes longer)
        assert p.submit(ref("square_chunk"), [5]).result(timeout=60) == [25]       # an early submit waits for the start
        assert len(p.worker_pids()) == 4 and p.last_start_ms is not None
        time.sleep(0.05)
        t0 = time.perf_counter()
        p.prestart()                                   # already running: nothing to do
        assert time.perf_counter() - t0 < 0.25 and len(p.worker_pids()) == 4
        # what is left on the critical path once the workers are up and hold the task file: one round trip per worker
        time.sleep(1.0)
        t0 = time.perf_counter()
        got = [f.result(timeout=60) for f in [p.submit(ref("sleep_chunk"), [k], (0.005,)) for k in range(4)]]
        took_ms = (time.perf_counter() - t0) * 1e3
        assert len({pid for ((_, pid),) in got}) == 4
        # (20 ms on an idle host; the bound leaves room for a contended one - a cold worker would add its interpreter start + numpy import)
        assert took_ms < 150.0, took_ms
    finally:
        p.close()


def test_an_empty_molecule_is_dropped_not_built_without_connectivity():
    """Advisor (round 5): `order_chunk` returned (None, None) for an empty molecule - "built, no connectivity" - and the stage
    then raised the every-molecule-or-none ValueError once the others supplied connectivities."""
    from ml_conformer_generator_amd import _rdkit_tasks as T
    assert T.order_chunk([([], np.zeros((0, 3)))], {}) == [None]
    from ml_conformer_generator_amd import rdkit_order as RO
    x, h, n = _batch(B=8)
    n[3] = 0
    import sys
    sys.path.insert(0, os.path.dirname(FAKE))
    import fake_host_tasks as F
    o, c, b = RO.batch_order_and_connectivity(lambda z, cc: F.order_chunk_no_sleep([(z, cc)])[0], x, h, n)
    assert b[3] is False and sum(b) == 6 and c[3].shape == (0, 0)          # molecule 5 is the Br one, molecule 3 the empty one
