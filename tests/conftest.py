import os
import sys

# The CPU oracle runs hundreds of small torch ops on 8 OpenMP threads.  libgomp's default wait policy spins ~300 000
# iterations at every barrier; on a host whose cores are shared (hypervisor steal, other jobs) a spinning thread burns the
# time slice the thread it waits for needs: the oracle sampler tests were measured 45x slower with six busy processes
# beside them (183 s instead of 4 s), 7x with a short spin (30 s).  Costs ~15 % on an idle host.  Must be set before
# libgomp is loaded, i.e. before `import torch`; an explicit setting in the environment wins.
os.environ.setdefault("GOMP_SPINCOUNT", "3000")

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The built library is git-ignored: a fresh checkout that runs the tests before `__graft_entry__.build()`
    gets it built here (hipcc cross-compiles gfx950 without a GPU).  Building is not a fallback: the product still
    refuses to run without the library."""
    lib = os.path.join(REPO, "ml_conformer_generator_amd", "libmlconfgen_hip.so")
    if not os.path.exists(lib) and not os.environ.get("MCG_LIB_PATH"):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(REPO, "ml_conformer_generator_amd", "csrc"), "-j4"], check=False)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: (torch.from_numpy(z[k]) if z[k].dtype.kind in "fi" and z[k].ndim > 0 else z[k]) for k in z.files}


_SD_CACHE = {}


def sd_for(g):
    """The synthetic EGNN weights a golden fixture was generated with (its `weight_seed` / `weight_recipe`)."""
    from ml_conformer_generator_amd.weights import synth_edm_state_dict
    key = (int(g["weight_seed"]), str(g["weight_recipe"]) if "weight_recipe" in g else "v2")
    if key not in _SD_CACHE:
        if key[1].startswith("gain"):          # the contractive legacy recipe: nn.Linear-family init x gain ("gain0.3")
            _SD_CACHE[key] = synth_edm_state_dict(key[0], weight_gain=float(key[1][4:]))
        else:
            _SD_CACHE[key] = synth_edm_state_dict(key[0], recipe=key[1])
    return _SD_CACHE[key]


@pytest.fixture(scope="session")
def edm_sd():
    return sd_for({"weight_seed": 1234, "weight_recipe": "v2"})


@pytest.fixture(scope="session")
def gcn_sd():
    from ml_conformer_generator_amd.weights import synth_adj_mat_seer_state_dict
    return synth_adj_mat_seer_state_dict(4321)


class TapeNoise:
    """Replays a recorded flat noise tape as successive `shape`-sized draws."""

    def __init__(self, flat, device="cpu"):
        self.flat = torch.as_tensor(flat, dtype=torch.float32)
        self.pos = 0
        self.device = device

    def __call__(self, shape):
        n = int(np.prod(shape))
        out = self.flat[self.pos:self.pos + n].reshape(shape)
        self.pos += n
        return out.to(self.device)
