"""Host fan-out of the RDKit-owned stages of `generate_conformers` ("RDKit MMFF fanned out over
host cores").

The reference canonicalises and standardises ONE molecule at a time on one Python thread
(`prepare_adj_mat_seer_input` -> `canonicalise`, utils/mol_utils.py:163-164; `standardize_mol` incl. up to 1 000 MMFF
iterations per molecule, conformer_generator.py:362-366, utils/standardizer.py:62-111).  With the sampler at 0.46 s per 64
molecules that thread - tens of ms per molecule - would decide "valid molecules/s" while the GPU and every other host core
idle.  `HostPool` runs those per-molecule functions in a pool of worker processes:

  * workers are FRESH interpreters (`subprocess` -> `_host_worker.py`, run by file path): never a `fork()` of this process
    (which owns a GPU context), never a re-exec of it; they import the task file (`_rdkit_tasks.py`: numpy + RDKit) and
    nothing of this package - no torch, no HIP library;
  * created lazily at the first submit - or ahead of it, in the background, by `prestart()` (the generator calls it BEFORE
    it launches the sampler, so the workers' start-up hides under the GPU) -, shared process-wide per (worker count,
    deadline) (`shared_pool`), shut down at exit; a worker that dies fails ITS task with `HostPoolError` and is replaced;
  * every task has a DEADLINE (`task_timeout_s`, default 60 s per chunk): RDKit is C++ - a pathological MMFF or
    kekulisation never returns to Python - and a stuck worker would park `generate_conformers` (and, sharded, the rank's
    place in the final collective) forever.  The serving thread waits for the reply with `select`; on expiry it kills
    THAT child by handle, replaces it, and the chunk's molecules come back as `None` = dropped - the reference's own
    "any failure => invalid" (utils/standardizer.py:108-109) - with a one-line warning, never an exception;
  * work is named by a `TaskRef` (file path + function name of a *chunk function* `f(items, *args) -> list`); results come
    back as `concurrent.futures.Future`s, so the caller consumes them in submission order while later chunks are still
    running - `MLConformerGenerator` launches the hand-off + GCN of one group of molecules as soon as ITS order results are
    in and submits that group's finish tasks behind it (order(g+1..) | GPU(g) | finish(..g) overlap);
  * an exception raised by a task (e.g. the reference's `ValueError` for a molecule without a perceived bond,
    utils/molgraph.py:152-155) is re-raised in the caller, same type, with the worker's traceback text attached
    (`.worker_traceback`); a `None` result keeps meaning "dropped".

`SerialExecutor` has the same `submit` and runs the chunk function in-process (`n_host_workers=0`): pooled and serial runs
execute the same code per molecule and return the same results in the same order (tests/test_host_pool.py).
"""
from __future__ import annotations

import atexit
import importlib.util
import os
import pickle
import queue
import select
import struct
import subprocess
import sys
import threading
import time
import warnings
from concurrent.futures import Future
from typing import Dict, List, Optional, Sequence, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
_WORKER = os.path.join(_HERE, "_host_worker.py")
RDKIT_TASKS = os.path.join(_HERE, "_rdkit_tasks.py")


DEFAULT_TASK_TIMEOUT_S = 60.0


class HostPoolError(RuntimeError):
    """A worker process died (or could not be started) while it held a task."""


class _TaskTimeout(Exception):
    """The worker did not answer within the task's deadline (internal: becomes `None` results + a warning)."""


class TaskRef:
    """A chunk function `f(items, *args) -> list` (one result per item) in a file a worker can load by path."""
    __slots__ = ("path", "func")

    def __init__(self, path: str, func: str):
        self.path, self.func = os.path.abspath(path), func

    def __repr__(self) -> str:
        return f"TaskRef({self.path!r}, {self.func!r})"

    def __eq__(self, other) -> bool:
        return isinstance(other, TaskRef) and (self.path, self.func) == (other.path, other.func)

    def __hash__(self) -> int:
        return hash((self.path, self.func))

    def load(self):
        """The function itself, loaded in THIS process (serial path)."""
        mod = _LOCAL.get(self.path)
        if mod is None:
            name = "_mcg_host_task_local_%d" % len(_LOCAL)
            spec = importlib.util.spec_from_file_location(name, self.path)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
            _LOCAL[self.path] = mod
        return getattr(mod, self.func)


_LOCAL: Dict[str, object] = {}

ORDER_TASK = TaskRef(RDKIT_TASKS, "order_chunk")
FINISH_TASK = TaskRef(RDKIT_TASKS, "finish_chunk")
SAMPLES_TASK = TaskRef(RDKIT_TASKS, "samples_chunk")


def default_workers() -> int:
    """`n_host_workers` default: min(32, this rank's share of the host cores) - MMFF is tens of ms per molecule, a generate
    call has 10..2 048.  A process that is already RESTRICTED to a subset of the cores (pinned by `affinity.pin`, a launcher, a
    cgroup cpuset) takes that subset as its share; otherwise, under a one-process-per-GPU launcher (`LOCAL_WORLD_SIZE`, set
    by torch.distributed.run), the cores are divided between the ranks of the node, like the intra-op thread cap of bench.py."""
    total = os.cpu_count() or 1
    try:
        allowed = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = total
    if 0 < allowed < total:
        return max(1, min(32, allowed))
    try:
        local = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        local = 1
    return max(1, min(32, total // local))


def chunk_bounds(n: int, size: int) -> List[Tuple[int, int]]:
    size = max(1, int(size))
    return [(lo, min(n, lo + size)) for lo in range(0, n, size)]


def task_chunk(n_items: int, n_workers: int, cap: int = 8) -> int:
    """Molecules per task: ~4 tasks per worker for load balance (MMFF times vary by an order of magnitude), at most `cap`
    so that one slow chunk is not the tail, at least 1."""
    if n_workers <= 0:
        return max(1, n_items)
    return max(1, min(cap, -(-n_items // (4 * n_workers))))


class SerialExecutor:
    """`submit` with the pool's signature, run in-process at submit time (no deadline: there is no one to enforce it)."""
    n_workers = 0

    def prestart(self, preload: Sequence = ()) -> None:
        return None

    def submit(self, ref: TaskRef, items: Sequence, args: tuple = (), timeout: Optional[float] = None) -> Future:
        fut: Future = Future()
        try:
            fut.set_result(ref.load()(list(items), *args))
        except BaseException as e:  # noqa: BLE001 - delivered where the pool would deliver it: at .result()
            fut.set_exception(e)
        return fut


class _Worker:
    def __init__(self, python: str, env: dict, preload: Sequence[str] = ()):
        r_task, w_task = os.pipe()
        r_res, w_res = os.pipe()
        try:
            # argv[3:] = task files the worker imports right away (numpy / RDKit load while the parent is still busy elsewhere)
            self.proc = subprocess.Popen([python, "-u", _WORKER, str(r_task), str(w_res)] + list(preload), pass_fds=(r_task, w_res),
                                         stdin=subprocess.DEVNULL, env=env, close_fds=True)
        except Exception:
            for fd in (r_task, w_task, r_res, w_res):
                os.close(fd)
            raise
        os.close(r_task)
        os.close(w_res)
        self.tx = os.fdopen(w_task, "wb")
        self.rx_fd = r_res                       # replies are length-prefixed frames read with a deadline (`_read_exact`)

    @property
    def pid(self) -> int:
        return self.proc.pid

    def _read_exact(self, n: int, deadline: Optional[float]) -> bytes:
        buf = bytearray()
        while len(buf) < n:
            if deadline is not None:
                left = deadline - time.monotonic()
                if left <= 0 or not select.select([self.rx_fd], [], [], left)[0]:
                    raise _TaskTimeout()
            chunk = os.read(self.rx_fd, min(1 << 20, n - len(buf)))
            if not chunk:
                raise EOFError("the worker closed its result pipe")
            buf += chunk
        return bytes(buf)

    def call(self, payload: bytes, timeout: Optional[float] = None):
        """One task round trip.  `timeout` (seconds, None = wait forever) covers the WHOLE reply - a worker that hangs in
        the middle of a frame is as stuck as one that never answers."""
        self.tx.write(payload)
        self.tx.flush()
        deadline = None if timeout is None else time.monotonic() + timeout
        (size,) = struct.unpack("<Q", self._read_exact(8, deadline))
        return pickle.loads(self._read_exact(size, deadline))

    def kill(self) -> None:
        """End THIS child now (by handle - never by name or pattern) and reap it."""
        try:
            self.proc.kill()
        except Exception:  # noqa: BLE001
            pass
        self.stop(timeout=2.0)

    def stop(self, timeout: float = 2.0) -> None:
        try:
            self.tx.close()                     # EOF on the task pipe = the worker's shutdown signal
        except Exception:  # noqa: BLE001
            pass
        if self.rx_fd is not None:
            try:
                os.close(self.rx_fd)
            except OSError:
                pass
            self.rx_fd = None
        try:
            self.proc.wait(timeout=timeout)
        except Exception:  # noqa: BLE001
            try:
                self.proc.kill()                # this exact child, by handle
                self.proc.wait(timeout=timeout)
            except Exception:  # noqa: BLE001
                pass


def _worker_env() -> dict:
    env = dict(os.environ)
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        env[k] = "1"                             # one core per worker: the pool IS the parallelism
    # a profiler's preloaded tool library would initialise the GPU inside every worker: workers are host-only
    for k in list(env):
        if k == "LD_PRELOAD" or k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_", "HSA_TOOLS_")):
            env.pop(k)
    return env


class HostPool:
    def __init__(self, n_workers: int, python: Optional[str] = None, task_timeout_s: Optional[float] = DEFAULT_TASK_TIMEOUT_S):
        if n_workers < 1:
            raise ValueError("HostPool needs at least one worker (0 workers = SerialExecutor)")
        if task_timeout_s is not None and not task_timeout_s > 0:
            raise ValueError("task_timeout_s must be positive (None = no deadline)")
        self.n_workers = int(n_workers)
        self.task_timeout_s = None if task_timeout_s is None else float(task_timeout_s)
        self._python = python or sys.executable
        self._q: "queue.Queue" = queue.Queue()
        self._threads: List[threading.Thread] = []
        self._workers: List[Optional[_Worker]] = []
        self._lock = threading.Lock()
        self._pid = None
        self._closed = False
        self._prestart: Optional[threading.Thread] = None
        self._preload: List[str] = []           # task files every new worker imports at start-up (`prestart(preload=)`)
        self.tasks_done = 0
        self.tasks_timed_out = 0
        self.workers_replaced = 0
        self.last_start_ms = None               # wall time of the last `start()` that really spawned the workers

    # ---------------------------------------------------------------- life cycle
    def start(self) -> "HostPool":
        with self._lock:
            if self._closed:
                raise HostPoolError("the pool is closed")
            if self._pid == os.getpid() and self._threads:
                return self
            # first use, or this is a forked copy of a process that had a pool: threads do not survive a fork
            t0 = time.perf_counter()
            self._q = queue.Queue()
            self._threads, self._workers = [], []
            env = _worker_env()
            try:
                for _ in range(self.n_workers):
                    self._workers.append(_Worker(self._python, env, self._preload))
            except Exception as e:
                for w in self._workers:
                    w.stop()
                self._workers = []
                raise HostPoolError(f"could not start a host worker: {e}") from e
            for k in range(self.n_workers):
                t = threading.Thread(target=self._serve, args=(k,), name=f"mcg-host-pool-{k}", daemon=True)
                t.start()
                self._threads.append(t)
            self._pid = os.getpid()
            self.last_start_ms = (time.perf_counter() - t0) * 1e3
        return self

    def prestart(self, preload: Sequence = ()) -> None:
        """Start the workers in the BACKGROUND and return at once: the caller (the generator, right before it launches the
        sampler) has seconds of GPU work ahead during which the host idles - spawning 32 interpreters costs ~0.3 s that
        would otherwise sit on the critical path of the first submit.  `preload`: `TaskRef`s (or file paths) whose task
        files the new workers import as they come up, so that the first task does not pay `import rdkit` / `import numpy`
        either.  A `submit` that comes early simply waits for the start in progress (same lock).  Start-up errors surface
        at that submit, not here."""
        with self._lock:
            for t in preload:
                path = t.path if isinstance(t, TaskRef) else os.path.abspath(str(t))
                if path not in self._preload:
                    self._preload.append(path)
            if self._closed or (self._pid == os.getpid() and self._threads):
                return
            if self._prestart is not None and self._prestart.is_alive():
                return

            def run():
                try:
                    self.start()
                except Exception:  # noqa: BLE001 - reported by the first submit (which calls start() again)
                    pass
            self._prestart = threading.Thread(target=run, name="mcg-host-pool-start", daemon=True)
            self._prestart.start()

    def worker_pids(self) -> List[int]:
        return [w.pid for w in self._workers if w is not None]

    def close(self) -> None:
        with self._lock:
            if self._closed:
                return
            self._closed = True
            threads, workers = self._threads, self._workers
            own = self._pid == os.getpid()
            self._threads, self._workers = [], []
        if not own:
            return
        for _ in threads:
            self._q.put(None)
        for t in threads:
            t.join(timeout=5.0)
        for w in workers:
            if w is not None:
                w.stop()

    def __enter__(self):
        return self.start()

    def __exit__(self, *exc):
        self.close()

    # ---------------------------------------------------------------- work
    def submit(self, ref: TaskRef, items: Sequence, args: tuple = (), timeout: Optional[float] = -1.0) -> Future:
        """One task = one chunk: `ref.func(list(items), *args)` in some worker -> Future of the result list.
        `timeout`: this task's deadline in seconds (default: the pool's `task_timeout_s`; None = none).  A task that
        misses it resolves to `[None] * len(items)` (every molecule of the chunk dropped) after its worker was killed and
        replaced - see the module docstring."""
        self.start()
        fut: Future = Future()
        items = list(items)
        payload = pickle.dumps((ref.path, ref.func, items, tuple(args)), protocol=pickle.HIGHEST_PROTOCOL)
        limit = self.task_timeout_s if (timeout is not None and timeout < 0) else timeout
        self._q.put((fut, payload, len(items), limit, f"{ref.func}"))
        return fut

    def _replace(self, k: int, w: Optional[_Worker], kill: bool) -> int:
        """Retire worker k (kill = it may still be running) and put a fresh one in its place; returns the old pid."""
        pid = w.pid if w is not None else -1
        if w is not None:
            if kill:
                w.kill()
            else:
                w.stop(timeout=0.5)
        with self._lock:
            if not self._closed and k < len(self._workers):
                try:
                    self._workers[k] = _Worker(self._python, _worker_env(), self._preload)
                    self.workers_replaced += 1
                except Exception:  # noqa: BLE001 - keep serving: the next task fails loudly too
                    self._workers[k] = None
        return pid

    def _serve(self, k: int) -> None:
        q = self._q
        while True:
            job = q.get()
            if job is None:
                return
            fut, payload, n_items, limit, what = job
            if not fut.set_running_or_notify_cancel():
                continue
            with self._lock:
                w = self._workers[k] if k < len(self._workers) else None
            try:
                if w is None:
                    raise EOFError("no worker")
                reply = w.call(payload, limit)
            except _TaskTimeout:
                # a stuck worker (RDKit is C++: it never comes back to Python): end THAT child, replace it BEFORE the caller
                # learns of it, and drop the chunk's molecules - the reference's "any failure => invalid"
                pid = self._replace(k, w, kill=True)
                self.tasks_timed_out += 1
                warnings.warn(f"ml_conformer_generator_amd.host_pool: {what} over {n_items} molecule(s) exceeded its {limit:g} s "
                              f"deadline in worker {pid}; the worker was replaced and the molecule(s) are dropped (None)",
                              RuntimeWarning, stacklevel=2)
                fut.set_result([None] * n_items)
                continue
            except BaseException as e:  # noqa: BLE001 - EOF / broken pipe (the worker died), a reply that does not unpickle here
                # (an exception class this process cannot import), MemoryError...: whatever it is, the pipe state is unknown -
                # recycle the worker and ALWAYS resolve the future (a serving thread that dies leaves `.result()` waiting forever)
                died = isinstance(e, (EOFError, OSError))
                pid = self._replace(k, w, kill=not died)
                why = "died while it held a task" if died else "returned a reply this process could not read and was recycled"
                fut.set_exception(HostPoolError(f"host worker {pid} {why} ({type(e).__name__}: {e})"))
                if not isinstance(e, Exception):
                    raise
                continue
            self.tasks_done += 1
            try:
                if reply[0]:
                    fut.set_result(reply[1])
                else:
                    exc = reply[1]
                    try:
                        exc.worker_traceback = reply[2]
                    except Exception:  # noqa: BLE001
                        pass
                    fut.set_exception(exc if isinstance(exc, BaseException) else HostPoolError(f"malformed worker reply: {exc!r}"))
            except Exception as e:  # noqa: BLE001 - a malformed reply must not take the serving thread down
                if not fut.done():
                    fut.set_exception(HostPoolError(f"malformed worker reply ({type(e).__name__}: {e})"))


_SHARED: Dict[Tuple[int, Optional[float]], HostPool] = {}
_SHARED_LOCK = threading.Lock()


def shared_pool(n_workers: int, task_timeout_s: Optional[float] = DEFAULT_TASK_TIMEOUT_S):
    """The process-wide pool with `n_workers` workers and this task deadline (created on first use, closed at exit);
    0 workers -> `SerialExecutor` (the calling thread: no deadline can be enforced there)."""
    if n_workers <= 0:
        return SerialExecutor()
    key = (int(n_workers), None if task_timeout_s is None else float(task_timeout_s))
    with _SHARED_LOCK:
        pool = _SHARED.get(key)
        if pool is None or pool._closed:
            pool = _SHARED[key] = HostPool(n_workers, task_timeout_s=task_timeout_s)
        return pool


@atexit.register
def _close_shared() -> None:
    with _SHARED_LOCK:
        pools = list(_SHARED.values())
        _SHARED.clear()
    for p in pools:
        try:
            p.close()
        except Exception:  # noqa: BLE001
            pass


def map_ordered(executor, ref: TaskRef, items: Sequence, args: tuple = (), chunk: Optional[int] = None) -> List:
    """All items through `ref`, results in item order (blocks until every chunk is done; the first failing chunk's
    exception is raised)."""
    items = list(items)
    if chunk is None:
        chunk = task_chunk(len(items), getattr(executor, "n_workers", 0))
    futs = [executor.submit(ref, items[lo:hi], args) for lo, hi in chunk_bounds(len(items), chunk)]
    out: List = []
    for f in futs:
        out.extend(f.result())
    return out
