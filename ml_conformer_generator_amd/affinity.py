"""CPU / NUMA placement of one rank process (one process per GPU).

N rank processes share one multi-socket host.  Each keeps a torch intra-op pool and a pool of host worker processes
(`host_pool.py`: RDKit's canonical order before the GCN, `redefine_bonds` + MMFF behind it); left to the scheduler, 8 ranks
x (16 torch threads + 32 workers) wander over both sockets and the launch thread of a rank may sit on the far side of the
GPU it feeds.  `rank_cpus` gives a rank the cores of ITS GPU's NUMA node - divided between the ranks whose GPUs share that
node - and `pin` applies them to every thread of the process; host-pool workers are children started afterwards and
inherit the mask.

Everything is read from sysfs WITHOUT a HIP / HSA call (a rank decides its placement before it touches the GPU, and
`bench.py`'s parent must not initialise one at all):

    /sys/class/kfd/kfd/topology/nodes/<k>/properties    simd_count > 0 = a GPU (KFD order = HIP order), drm_render_minor
    /sys/class/drm/renderD<minor>/device/numa_node      the GPU's NUMA node (-1 = unknown)
    /sys/devices/system/node/node<k>/cpulist            the node's cores

`root` re-bases every path (tests/test_host_logic.py builds a fake tree).  Anything missing -> the allowed cores are
split evenly by local rank (never an error: placement is an optimisation).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

_VISIBLE_VARS = ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")


def parse_cpulist(text: str) -> List[int]:
    """"0-3,8,10-11" -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    out: List[int] = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return sorted(set(out))


def _read(path: str) -> Optional[str]:
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def gpu_render_minors(root: str = "/") -> List[int]:
    """DRM render minors of the GPUs in KFD topology order (= the HIP device order with no *_VISIBLE_DEVICES set)."""
    base = os.path.join(root, "sys/class/kfd/kfd/topology/nodes")
    try:
        nodes = sorted((d for d in os.listdir(base) if d.isdigit()), key=int)
    except OSError:
        return []
    minors = []
    for d in nodes:
        text = _read(os.path.join(base, d, "properties"))
        if text is None:
            continue
        props: Dict[str, str] = {}
        for ln in text.splitlines():
            k, _, v = ln.partition(" ")
            props[k] = v.strip()
        try:
            if int(props.get("simd_count", "0")) > 0:
                minors.append(int(props.get("drm_render_minor", "-1")))
        except ValueError:
            pass
    return minors


def visible_gpu_minors(root: str = "/", env: Optional[dict] = None) -> List[int]:
    """`gpu_render_minors` narrowed / re-ordered by ROCR_ / HIP_ / CUDA_VISIBLE_DEVICES (integer lists; applied in that
    order, each indexing the list the previous one left - ROCr filters first, HIP then filters what ROCr shows)."""
    env = os.environ if env is None else env
    minors = gpu_render_minors(root)
    for var in _VISIBLE_VARS:
        v = env.get(var)
        if v is None:
            continue
        try:
            idx = [int(x) for x in v.split(",") if x.strip() != ""]
        except ValueError:           # UUID form: not resolvable from sysfs alone - leave the list alone
            continue
        minors = [minors[i] for i in idx if 0 <= i < len(minors)]
    return minors


def gpu_numa_node(device_index: int, root: str = "/", env: Optional[dict] = None) -> int:
    """NUMA node of HIP device `device_index`; -1 when sysfs does not say."""
    minors = visible_gpu_minors(root, env)
    if not 0 <= device_index < len(minors) or minors[device_index] < 0:
        return -1
    text = _read(os.path.join(root, "sys/class/drm", "renderD%d" % minors[device_index], "device/numa_node"))
    try:
        return int(text.strip()) if text is not None else -1
    except ValueError:
        return -1


def numa_cpus(node: int, root: str = "/") -> List[int]:
    text = _read(os.path.join(root, "sys/devices/system/node", "node%d" % node, "cpulist"))
    try:
        return parse_cpulist(text) if text is not None else []
    except ValueError:
        return []


def _share(cpus: Sequence[int], k: int, n: int) -> List[int]:
    """The k-th of n contiguous shares of `cpus` (everything when there are fewer cores than sharers)."""
    cpus = list(cpus)
    if n <= 1 or len(cpus) < n:
        return cpus
    base, extra = divmod(len(cpus), n)
    lo = k * base + min(k, extra)
    return cpus[lo: lo + base + (1 if k < extra else 0)]


def rank_cpus(local_rank: int, local_world: int, device_index: Optional[int] = None, root: str = "/",
              allowed: Optional[Sequence[int]] = None, env: Optional[dict] = None) -> List[int]:
    """The cores rank `local_rank` of `local_world` ranks on this host should run on: the cores of its GPU's NUMA node that
    this process is allowed to use, divided between the local ranks whose GPUs sit on the same node (rank r drives device
    r % n_visible, the launcher convention; `device_index` overrides it for THIS rank).  Unknown topology: an even split
    of `allowed` by local rank."""
    if allowed is None:
        try:
            allowed = sorted(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            allowed = list(range(os.cpu_count() or 1))
    allowed = sorted(allowed)
    local_world = max(1, int(local_world))
    dev = local_rank if device_index is None else device_index
    node = gpu_numa_node(dev, root, env)
    cpus = [c for c in numa_cpus(node, root) if c in set(allowed)] if node >= 0 else []
    if not cpus:
        return _share(allowed, local_rank % local_world, local_world)
    # the local ranks whose GPUs sit on the same NUMA node, in rank order (rank r drives device r % n_visible)
    n_vis = max(1, len(visible_gpu_minors(root, env)))
    same = [r for r in range(local_world) if r == local_rank or gpu_numa_node(r % n_vis, root, env) == node]
    return _share(cpus, same.index(local_rank), len(same))


def pin(cpus: Sequence[int]) -> List[int]:
    """Restrict EVERY thread of this process to `cpus` (Linux `sched_setaffinity` is per thread: pid 0 would move the calling
    thread only and leave an already-started torch / OpenMP pool where it was); threads and child processes created later
    inherit the mask.  Returns the mask now in force for the calling thread ([] where the platform has no affinity calls)."""
    cpus = sorted(set(int(c) for c in cpus))
    if not cpus or not hasattr(os, "sched_setaffinity"):
        return []
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = [0]
    for tid in tids:
        try:
            os.sched_setaffinity(tid, cpus)
        except OSError:              # a thread that exited meanwhile, or a core outside the cgroup's set
            pass
    try:
        return sorted(os.sched_getaffinity(0))
    except OSError:
        return []


def pin_rank(local_rank: Optional[int] = None, local_world: Optional[int] = None, device_index: Optional[int] = None,
             root: str = "/") -> List[int]:
    """`pin(rank_cpus(...))` with the launcher's environment as the default (`LOCAL_RANK`, `LOCAL_WORLD_SIZE` of
    torch.distributed.run; a lone process = rank 0 of 1)."""
    def _env_int(name, default):
        try:
            return int(os.environ.get(name, default))
        except ValueError:
            return default
    lr = _env_int("LOCAL_RANK", 0) if local_rank is None else local_rank
    lw = _env_int("LOCAL_WORLD_SIZE", _env_int("WORLD_SIZE", 1)) if local_world is None else local_world
    return pin(rank_cpus(lr, lw, device_index, root))
