"""The C-ABI library loads (no GPU needed) and exports every symbol include/mlconfgen_hip.h declares;
the product fails loudly when the library is missing."""
import ctypes
import os
import re

import pytest

from conftest import REPO
from ml_conformer_generator_amd import _lib

HEADER = os.path.join(REPO, "include", "mlconfgen_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mcg_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    syms = declared_symbols()
    assert "mcg_egnn_dynamics" in syms and "mcg_gcn_forward" in syms and len(syms) >= 18
    assert sorted(_lib.SIGNATURES) == syms


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "build first: python -c 'import __graft_entry__ as g; g.build()'"
    h = _lib.lib()
    for s in declared_symbols():
        assert hasattr(h, s), s
    assert h.mcg_abi_version() == 4
    assert h.mcg_last_error() is not None


def test_argument_errors_do_not_need_a_gpu():
    h = _lib.lib()
    out = ctypes.c_void_p()
    assert h.mcg_plan_create(0, 5, None, 0, ctypes.byref(out)) != 0           # bad B / null sizes
    assert b"mcg_plan_create" in h.mcg_last_error()
    assert h.mcg_egnn_create(None, 0, 420, 9, ctypes.byref(out)) != 0
    assert h.mcg_gcn_create(None, 22, ctypes.byref(out)) != 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", os.path.join(REPO, "does_not_exist.so"))
    with pytest.raises(_lib.McgError, match="no CPU fallback"):
        _lib.lib()
    from ml_conformer_generator_amd import MLConformerGenerator
    with pytest.raises(_lib.McgError):
        MLConformerGenerator(edm_weights={}, adj_mat_seer_weights={})


def test_cpu_device_is_rejected():
    import torch
    from ml_conformer_generator_amd import MLConformerGenerator
    with pytest.raises(ValueError, match="no CPU fallback"):
        MLConformerGenerator(device=torch.device("cpu"), edm_weights={}, adj_mat_seer_weights={})


def test_plan_size_limit_is_reported_not_overflowed():
    """32-bit offsets inside the kernels: a plan beyond 1e6 atoms is refused with a message (no GPU work: the check
    precedes every allocation)."""
    import ctypes as C
    import numpy as np
    from ml_conformer_generator_amd import _lib
    L = _lib.lib()
    n = np.full(30000, 39, dtype=np.int32)                      # 1.17e6 atoms
    h = C.c_void_p()
    rc = L.mcg_plan_create(30000, 39, n.ctypes.data_as(C.c_void_p), 0, C.byref(h))
    assert rc != 0 and b"too large" in L.mcg_last_error()


def test_header_is_plain_c_and_argument_checks_of_the_new_entry_points(tmp_path):
    """`include/mlconfgen_hip.h` is the drop-in boundary: it must compile as C99 and as C++ on its own (plain pointers and
    sizes, no torch / HIP types), and the round-4 entry points refuse bad arguments without a GPU."""
    import shutil
    import subprocess
    src = tmp_path / "h.c"
    src.write_text('#include "mlconfgen_hip.h"\nint main(void) { return mcg_abi_version() == 0; }\n')
    inc = os.path.join(REPO, "include")
    if shutil.which("gcc"):
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-I", inc, str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    if shutil.which("g++"):
        r = subprocess.run(["g++", "-std=c++11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c++", "-I", inc, str(src)],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    h = _lib.lib()
    assert h.mcg_handoff_ex(None, None, None, 4, 19, 1.3, None, None, None, None, None, None, None, None) != 0
    assert b"mcg_handoff_ex" in h.mcg_last_error()
    counts = (ctypes.c_int64 * 32)()
    assert h.mcg_debug_gemm_launches(counts, 1) == 0 and all(v >= 0 for v in counts)
