"""ctypes binding of libmlconfgen_hip.so (the C ABI in include/mlconfgen_hip.h).

The library is the product: if it is missing or does not export a declared
symbol, importing this module's `lib()` raises - there is no CPU/PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Sequence

_HERE = os.path.dirname(os.path.abspath(__file__))
# MCG_LIB_PATH: measurement builds only (tools/build_variants.sh: ablation / experiment variants of the same library)
LIB_PATH = os.environ.get("MCG_LIB_PATH") or os.path.join(_HERE, "libmlconfgen_hip.so")

_vp, _i, _f, _d = C.c_void_p, C.c_int, C.c_float, C.c_double
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); mirrors include/mlconfgen_hip.h one to one
SIGNATURES: Dict[str, tuple] = {
    "mcg_last_error": (C.c_char_p, []),
    "mcg_abi_version": (_i, []),
    "mcg_egnn_create": (_i, [_pp, _i, _i, _i, _pp]),
    "mcg_egnn_destroy": (None, [_vp]),
    "mcg_egnn_set_precision": (_i, [_vp, _i]),
    "mcg_egnn_set_option": (_i, [_vp, _i, _i]),
    "mcg_debug_gemm_launches": (_i, [_vp, _i]),
    "mcg_plan_create": (_i, [_i, _i, _vp, _i, _pp]),
    "mcg_plan_create_ex": (_i, [_i, _i, _vp, _vp, _pp]),
    "mcg_plan_destroy": (None, [_vp]),
    "mcg_pool_stats": (_i, [_vp, _i]),
    "mcg_plan_info": (_i, [_vp, _vp]),
    "mcg_plan_ranges": (_i, [_vp]),
    "mcg_plan_set_latency_mode": (_i, [_vp, _i]),
    "mcg_plan_check_tables": (_i, [_i, _i, _vp, _vp, _i, _vp]),
    "mcg_egnn_dynamics": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mcg_egnn_block_debug": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "mcg_egnn_gcl_debug": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "mcg_plan_peek": (_i, [_vp, _i, _vp, _vp]),
    "mcg_bench_edge": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "mcg_bench_edge_incall": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp]),
    "mcg_sampler_noise": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "mcg_sampler_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _vp, _vp]),
    "mcg_sampler_decode": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "mcg_sampler_blend": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _f, _f, _f, _i, _vp]),
    "mcg_egnn_aggregate": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _vp]),
    "mcg_gcn_create": (_i, [_pp, _i, _pp]),
    "mcg_gcn_destroy": (None, [_vp]),
    "mcg_gcn_forward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
    "mcg_gcn_check": (_i, [_vp]),
    "mcg_shape_tanimoto": (_i, [_vp, _i, _vp, _vp, _i, _i, _vp, _i, _vp, _i, _f, _f, _vp, _vp, _vp]),
    "mcg_handoff": (_i, [_vp, _vp, _vp, _i, _i, _d, _vp, _vp, _vp, _vp]),
    "mcg_handoff_ex": (_i, [_vp, _vp, _vp, _i, _i, _d, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mcg_bond_writeback": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp]),
    "mcg_ifm_merge": (_i, [_vp, _vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _vp, _vp, _vp]),
}



class PlanOpts(C.Structure):
    """`mcg_plan_opts` (include/mlconfgen_hip.h): zero-initialised = the library's defaults."""
    _fields_ = [("edge_mt", C.c_int32), ("n_ranges", C.c_int32), ("four_tile_units", C.c_int32), ("reserved", C.c_int32 * 5)]


ALL_FOUR_TILE = 0x3fffffff          # MCG_ALL_FOUR_TILE
OPT_X6_GEMM, OPT_GEMM_RN, OPT_GEMM_X6_RN, OPT_GEMM_BF16_LDS, OPT_NODE_FUSED = 1, 2, 3, 4, 5

_lib = None


class McgError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raise if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise McgError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C ml_conformer_generator_amd/csrc`. There is no CPU fallback for the hot path.")
    import torch  # noqa: F401  - loads the process's HIP runtime first (same SONAME as ours)
    handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as e:
            raise McgError(f"{LIB_PATH} does not export `{name}` (stale build?)") from e
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return handle


def check(code: int, what: str) -> None:
    if code != 0:
        msg = lib().mcg_last_error()
        raise McgError(f"{what} failed (code {code}): {msg.decode() if msg else ''}")


def host_ptr_array(tensors: Sequence) -> "C.Array":
    """void*[n] of HOST data pointers of contiguous fp32 CPU tensors."""
    arr = (C.c_void_p * len(tensors))()
    for k, t in enumerate(tensors):
        assert t.device.type == "cpu" and t.is_contiguous() and str(t.dtype) == "torch.float32"
        arr[k] = t.data_ptr()
    return arr


def dptr(t) -> int:
    """Device (or host) data pointer of a contiguous tensor, as int for ctypes."""
    if t is None:
        return None
    assert t.is_contiguous(), "kernel operands must be contiguous"
    return t.data_ptr()


def current_stream_ptr(device) -> int:
    import torch
    return torch.cuda.current_stream(device).cuda_stream
