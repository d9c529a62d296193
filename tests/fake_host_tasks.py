"""Chunk functions for tests/test_host_pool.py: stand-ins for the RDKit stages that a `host_pool` worker loads BY FILE PATH
(plain numpy, deterministic functions of their inputs; RDKit exists neither here nor on the GPU boxes)."""
import os
import sys
import time

import numpy as np

PER_MOLECULE_S = 0.002          # the review's "2 ms-per-molecule fake"


def _order_one(z, coords):
    """(order, connectivity) from geometry alone: atoms by distance from the centroid (ties by index), bonds < 1.9 A."""
    c = np.asarray(coords, dtype=np.float64)
    n = len(z)
    r = np.linalg.norm(c - c.mean(0, keepdims=True), axis=1)
    order = sorted(range(n), key=lambda i: (round(float(r[i]), 9), i))
    d = np.linalg.norm(c[:, None, :] - c[None, :, :], axis=2)
    return order, ((d < 1.9) & ~np.eye(n, dtype=bool)).astype(np.uint8)


def order_chunk(items):
    out = []
    for z, c in items:
        time.sleep(PER_MOLECULE_S)
        if len(z) > 0 and z[0] == 35:
            out.append(None)                                   # "MolFromXYZBlock returned None": dropped downstream
        else:
            out.append(_order_one(z, c))
    return out


def order_chunk_no_sleep(items):
    return [None if (len(z) > 0 and z[0] == 35) else _order_one(z, c) for z, c in items]


def order_chunk_bondless_raises(items):
    """The reference raises ValueError for a molecule without a perceived bond (utils/molgraph.py:152-155)."""
    out = []
    for z, c in items:
        o, conn = _order_one(z, c)
        if conn.sum() == 0:
            raise ValueError("Bonds must be specified for the molecule - no connectivity perceived.")
        out.append((o, conn))
    return out


def order_chunk_half_connectivity(items):
    return [(_order_one(z, c)[0], _order_one(z, c)[1] if len(z) % 2 else None) for z, c in items]


def finish_chunk(items, optimise_geometry):
    """A picklable summary per molecule; None (= dropped) for a molecule without bonds."""
    out = []
    for z, coords, bonds in items:
        time.sleep(PER_MOLECULE_S)
        nb = sum(1 for i, row in enumerate(bonds) for j in range(i) if row[j] != 0)
        if nb == 0:
            out.append(None)
        else:
            out.append({"z": tuple(int(v) for v in z), "xyz": "%.9f" % float(np.asarray(coords, dtype=np.float64).sum()),
                        "bonds": nb, "mmff": bool(optimise_geometry)})
    return out


def sleep_chunk(items, seconds):
    for _ in items:
        time.sleep(seconds)
    return [(i, os.getpid()) for i in items]


def square_chunk(items):
    return [None if i % 7 == 3 else i * i for i in items]


def raise_on_13(items):
    for i in items:
        if i == 13:
            raise ValueError("thirteen is not a molecule")
    return list(items)


def crash_chunk(items):
    if 5 in items:
        os._exit(3)
    return list(items)


def introspect_chunk(items):
    """What a worker process is: its pid / parent, its argv and which heavy modules it holds."""
    heavy = sorted(m for m in sys.modules if m.split(".")[0] in ("torch", "ml_conformer_generator_amd"))
    return [{"pid": os.getpid(), "ppid": os.getppid(), "argv0": sys.argv[0], "heavy": heavy,
             "omp": os.environ.get("OMP_NUM_THREADS")} for _ in items]


def hang_on_7(items):
    """A task that never returns for one of its molecules (RDKit is C++: a pathological MMFF does not come back)."""
    if 7 in items:
        while True:
            time.sleep(3600)
    return [i + 100 for i in items]


class TaskLocalError(Exception):
    """Defined in a file the worker loads BY PATH (module `_mcg_host_task_<n>`): the parent cannot import it."""


def raise_task_local(items):
    raise TaskLocalError("only the worker knows this class")


def sys_path_chunk(items):
    return [list(sys.path) for _ in items]


def finish_tag_chunk(items, optimise_geometry):
    """A finished 'molecule' per item (never dropped): what travels through `distributed.gather_objects`."""
    return [("mol", len(z), bool(optimise_geometry)) for z, _, _ in items]


def finish_raises_chunk(items, optimise_geometry):
    raise RuntimeError("finisher fault injected")
