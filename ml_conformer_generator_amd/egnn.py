"""EGNN denoiser behind the reference's operator seam (egnn.py:448-541 `EGNNDynamics`).

`EGNNDynamics.forward(t, xh, node_mask, edge_mask, context)` keeps the 5-tensor
contract of the reference (and of its ONNX export, onnx_export_utils.py:38-49);
the arithmetic runs in libmlconfgen_hip.so.  Weights are handed over in the
reference's state-dict layout and repacked by the library.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Dict, Optional

import torch

from . import _lib
from .config import EGNN_HIDDEN, EGNN_N_BLOCKS
from .weights import edm_spec


class BatchPlan:
    """Batch geometry (molecule sizes -> compact node/edge tiling) + device workspace.
    Replaces the reference's per-call `get_adj_matrix` and mask tensors."""

    def __init__(self, n_nodes: torch.Tensor, max_n_nodes: int, device: torch.device, edge_mt: int = 0,
                 n_ranges: int = 0, four_tile_units: int = 0):
        """`edge_mt`, `n_ranges`, `four_tile_units`: the fields of `mcg_plan_opts` (0 = the library's choice)."""
        L = _lib.lib()
        n_host = n_nodes.detach().to("cpu", torch.int32).contiguous().reshape(-1)
        self.B = int(n_host.numel())
        self.N = int(max_n_nodes)
        self.device = torch.device(device)
        self.n_nodes_host = n_host
        self._h = C.c_void_p()
        opts = _lib.PlanOpts(edge_mt=int(edge_mt), n_ranges=int(n_ranges), four_tile_units=int(four_tile_units))
        with torch.cuda.device(self.device):
            _lib.check(L.mcg_plan_create_ex(self.B, self.N, n_host.data_ptr(), C.byref(opts), C.byref(self._h)),
                       "mcg_plan_create_ex")
        info = torch.zeros(8, dtype=torch.int32)
        _lib.check(L.mcg_plan_info(self._h, info.data_ptr()), "mcg_plan_info")
        (self.n_real_nodes, self.n_real_edges, self.edge_mt, self.n_edge_waves, self.n_pslots, _, _,
         self.n_edge_tiles) = [int(v) for v in info]
        self.n_ranges = int(L.mcg_plan_ranges(self._h))      # molecule ranges (HIP streams) the plan really runs

    @property
    def handle(self):
        return self._h

    def set_latency_mode(self, mode: int) -> None:
        """-1 auto (four-tile workgroups for complete rounds of the chip, quarter-tile ones for the rest), 0 four-tile
        workgroups only, 1 the stand-alone column-split kernel with per-wave partial sums (mcg_plan_set_latency_mode)."""
        _lib.check(_lib.lib().mcg_plan_set_latency_mode(self._h, int(mode)), "mcg_plan_set_latency_mode")

    def node_mask(self) -> torch.Tensor:
        idx = torch.arange(self.N).unsqueeze(0)
        return (idx < self.n_nodes_host.unsqueeze(1)).to(torch.float32).unsqueeze(2).to(self.device)

    def edge_mask(self) -> torch.Tensor:
        """The ONE edge mask the plan's kernels implement, [B*N*N, 1] on the device: outer product of the prefix node
        mask with the diagonal removed (`prepare_masks`, mol_utils.py:246-249).  Built once per plan."""
        em = getattr(self, "_edge_mask", None)
        if em is None:
            nm = self.node_mask().squeeze(2)
            em = nm.unsqueeze(1) * nm.unsqueeze(2) * (1.0 - torch.eye(self.N, device=self.device)).unsqueeze(0)
            em = self._edge_mask = em.reshape(self.B * self.N * self.N, 1)
        return em

    def __del__(self):
        try:
            if self._h:
                _lib.lib().mcg_plan_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass


def sizes_from_node_mask(node_mask: torch.Tensor) -> torch.Tensor:
    """n_nodes[B] from a [B,N,1] (or [B,N]) prefix mask; rejects non-prefix masks (the
    reference only ever builds prefix masks, mol_utils.py:241-243)."""
    nm = node_mask.reshape(node_mask.shape[0], -1)
    n = nm.sum(1).round().to(torch.int64)
    expect = (torch.arange(nm.shape[1], device=nm.device).unsqueeze(0) < n.unsqueeze(1)).to(nm.dtype)
    if not bool(torch.equal(nm, expect)):
        raise ValueError("node_mask must be a 0/1 prefix mask per sample (as built by prepare_masks)")
    return n.to("cpu", torch.int32)


class EGNNDynamics(torch.nn.Module):
    def __init__(self, in_node_nf: int = 9, context_node_nf: int = 3, n_dims: int = 3,
                 hidden_nf: int = EGNN_HIDDEN, device: torch.device = torch.device("cuda:0"),
                 normalization_factor: float = 100.0, n_blocks: int = EGNN_N_BLOCKS):
        super().__init__()
        if hidden_nf != EGNN_HIDDEN or in_node_nf != 9 or context_node_nf != 3 or n_dims != 3 \
                or normalization_factor != 100.0:
            raise ValueError("the HIP kernels are specialised for the published architecture "
                             "(in_node_nf=9, context_node_nf=3, hidden_nf=420, normalization 100)")
        self.in_node_nf, self.context_node_nf, self.n_dims = in_node_nf, context_node_nf, n_dims
        self.hidden_nf, self.n_blocks = hidden_nf, n_blocks
        self.device = torch.device(device)
        self._h = C.c_void_p()
        self.compute_dtype = "f32"
        self._plans: Dict[tuple, BatchPlan] = {}      # LRU (dict order = recency), `plan_cache_size` entries
        self.plan_cache_size = 8
        self._edge_mask_ok = None          # (weakref to the last verified edge_mask tensor, its version, its plan)
        self._node_mask_ok = None          # (weakref to the last verified node_mask tensor, its version, its sizes)

    # -- weights ------------------------------------------------------------------
    def load_reference_state_dict(self, sd: Dict[str, torch.Tensor], prefix: str = "dynamics.egnn.") -> None:
        """Accepts the keys of the reference checkpoint."""
        L = _lib.lib()
        spec = edm_spec(self.hidden_nf, self.in_node_nf + self.context_node_nf, self.n_blocks)
        tensors = []
        for key, shape, _, _ in spec:
            k = prefix + key[len("dynamics.egnn."):]
            if k not in sd:
                raise RuntimeError(f"Missing key(s) in state_dict: \"{k}\"")
            t = sd[k].detach().to("cpu", torch.float32).contiguous()
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: {tuple(t.shape)} vs {tuple(shape)}")
            tensors.append(t)
        arr = _lib.host_ptr_array(tensors)
        # cached plans hold HIP graphs with the old model's device weight pointers baked in: drop them with it
        self._plans.clear()
        if self._h:
            L.mcg_egnn_destroy(self._h)
            self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.mcg_egnn_create(arr, len(tensors), self.hidden_nf, self.n_blocks, C.byref(self._h)),
                       "mcg_egnn_create")
        if self.compute_dtype != "f32":          # a fresh C model starts in exact fp32: keep the mode the caller chose
            dtype, self.compute_dtype = self.compute_dtype, "f32"
            self.set_precision(dtype)

    @property
    def handle(self):
        if not self._h:
            raise _lib.McgError("EGNNDynamics has no weights loaded")
        return self._h

    def set_precision(self, compute_dtype: str = "f32") -> None:
        """"f32" (default): exact fp32 MFMA.  "bf16": MFMA operands rounded to bf16, fp32 accumulate,
        fp32 coordinates/aggregates/epilogues (BASELINE.json configs[4]).  "f32x6": fp32-accurate edge-MLP
        contraction as six bf16 partial products of three-part fp32 operands (see include/mlconfgen_hip.h)."""
        if compute_dtype not in ("f32", "bf16", "f32x6"):
            raise ValueError("compute_dtype must be 'f32', 'bf16' or 'f32x6'")
        _lib.check(_lib.lib().mcg_egnn_set_precision(self.handle, {"f32": 0, "bf16": 1, "f32x6": 2}[compute_dtype]),
                   "mcg_egnn_set_precision")
        if compute_dtype != self.compute_dtype:
            self._plans.clear()          # tilings differ between the precisions
        self.compute_dtype = compute_dtype

    def set_option(self, option: int, value: int) -> None:
        """Model-level measurement options (`mcg_egnn_set_option`: _lib.OPT_X6_GEMM / OPT_GEMM_RN / OPT_GEMM_X6_RN)."""
        _lib.check(_lib.lib().mcg_egnn_set_option(self.handle, int(option), int(value)), "mcg_egnn_set_option")
        self._plans.clear()          # captured graphs hold the old launch configuration

    # -- plans --------------------------------------------------------------------
    def plan(self, n_nodes: torch.Tensor, max_n_nodes: int, edge_mt: int = 0, four_tile_units: int = 0,
             n_ranges: int = 0) -> BatchPlan:
        """Cached plan for a batch of molecule sizes.  `edge_mt`, `four_tile_units`, `n_ranges`: `mcg_plan_opts`
        (0 = the library's choice; tests and measurement tools pass them explicitly)."""
        if edge_mt == 0 and self.compute_dtype in ("bf16", "f32x6") and int(n_nodes.min()) >= 6:
            edge_mt = 4          # 64-row workgroup tiles (needs <= 16 nodes per 64 edge rows)
        # split-operand mode: two molecule ranges on two streams from ~1 500 edge tiles on (its edge kernel is short
        # against the node phase, which the other range's edge kernel then overlaps: 3.53 -> 3.18 ms at config 2)
        if n_ranges == 0 and edge_mt == 4 and self.compute_dtype == "f32x6":
            nn = n_nodes.reshape(-1).to(torch.long)
            n_ranges = 2 if int((nn * (nn - 1)).sum()) >= 1500 * 16 else 1
        key = (int(max_n_nodes), int(edge_mt), int(four_tile_units), int(n_ranges), tuple(int(v) for v in n_nodes.reshape(-1).tolist()))
        p = self._plans.get(key)
        if p is not None:
            self._plans[key] = self._plans.pop(key)          # least-recently-USED goes first: a hit moves to the back
            return p
        # A ragged caller meets a new size vector on every call: the oldest plan goes (its device blocks return to the
        # library's plan pool and are handed to the new plan - no driver allocation in steady state).
        while len(self._plans) >= max(1, self.plan_cache_size):
            self._plans.pop(next(iter(self._plans)))
        p = BatchPlan(n_nodes, max_n_nodes, self.device, edge_mt, n_ranges, four_tile_units)
        self._plans[key] = p
        return p

    def sizes_for(self, node_mask: torch.Tensor) -> torch.Tensor:
        """n_nodes[B] (CPU int32) of a prefix node mask WITHOUT a device compare + host sync where the verdict is already
        known: masks built by this package carry their sizes (`mol_utils.tag_canonical_masks`), and the last externally
        built mask is remembered by tensor identity + version counter - the reference's own sampler loop passes the same
        `node_mask` tensor to `dynamics` on every step (equivariant_diffusion.py:187) and pays the check once."""
        tag = getattr(node_mask, "_mcg_mask_tag", None)
        if tag is not None and tag[1] == node_mask._version:
            return tag[2]
        hit = self._node_mask_ok
        if hit is not None and hit[0]() is node_mask and hit[1] == node_mask._version:
            return hit[2]
        sizes = sizes_from_node_mask(node_mask)
        self._node_mask_ok = (weakref.ref(node_mask), node_mask._version, sizes)
        return sizes

    def release_cached_memory(self) -> dict:
        """Drop what this object and the library cache on the device: the LRU plan cache (their graphs, tables and
        workspaces) and the free blocks of the library's plan pool (up to 4 GiB per device that PyTorch's caching allocator
        cannot see - call this from an out-of-memory retry path).  Returns the pool's counters after the trim."""
        self._plans.clear()
        self._edge_mask_ok = self._node_mask_ok = None
        stats = torch.zeros(4, dtype=torch.int64)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().mcg_pool_stats(stats.data_ptr(), 1), "mcg_pool_stats")
        return dict(zip(("in_use_bytes", "cached_bytes", "driver_allocs", "pool_hits"), (int(v) for v in stats)))

    def check_edge_mask(self, plan: BatchPlan, edge_mask, node_mask=None) -> None:
        """The reference multiplies every message by whatever `edge_mask` the caller passes (egnn.py:477-478,51,127); the
        HIP kernels implement exactly one - the canonical mask of the plan's prefix node mask.  Any other mask is REFUSED
        (ValueError) instead of being silently ignored.  One device compare + sync per distinct mask tensor: the verdict is
        cached on the tensor object + its version counter, so the reference's sampler loop (the same tensor every step)
        pays it once."""
        if edge_mask is None:
            return
        tag, ntag = getattr(edge_mask, "_mcg_mask_tag", None), getattr(node_mask, "_mcg_mask_tag", None)
        if (tag is not None and ntag is not None and tag[0] is ntag[0] and tag[1] == edge_mask._version
                and ntag[1] == node_mask._version):
            return                          # built together by prepare_masks, untouched since: canonical by construction
        hit = self._edge_mask_ok
        if hit is not None and hit[0]() is edge_mask and hit[1] == edge_mask._version and hit[2]() is plan:
            return
        if edge_mask.numel() != plan.B * plan.N * plan.N:
            raise ValueError(f"edge_mask has {edge_mask.numel()} entries, expected B*N*N = {plan.B * plan.N * plan.N}")
        em = edge_mask.reshape(-1, 1).to(self.device, torch.float32)
        if not bool(torch.equal(em, plan.edge_mask())):
            raise ValueError("edge_mask is not the canonical mask of node_mask (outer product minus the diagonal, as built "
                             "by prepare_masks): the HIP denoiser supports prefix node masks with that edge mask only")
        self._edge_mask_ok = (weakref.ref(edge_mask), edge_mask._version, weakref.ref(plan))

    # -- operator seam ------------------------------------------------------------
    def run(self, plan: BatchPlan, t: torch.Tensor, xh: torch.Tensor, context: torch.Tensor,
            out: Optional[torch.Tensor] = None) -> torch.Tensor:
        L = _lib.lib()
        if out is None:
            out = torch.empty_like(xh)
        _lib.check(L.mcg_egnn_dynamics(self.handle, plan.handle, _lib.dptr(t), _lib.dptr(xh), _lib.dptr(context),
                                       _lib.dptr(out), _lib.current_stream_ptr(self.device)), "mcg_egnn_dynamics")
        return out

    @torch.no_grad()
    def forward(self, t, xh, node_mask, edge_mask, context):
        """out[B,N,11] - same contract as the reference (egnn.py:472-513).  `edge_mask` must be the canonical mask of
        `node_mask` (it is fully determined by it in every caller of the reference); anything else raises ValueError
        (`check_edge_mask`).  None skips the check."""
        B, N, _ = xh.shape
        plan = self.plan(self.sizes_for(node_mask), N)
        self.check_edge_mask(plan, edge_mask, node_mask)
        f32 = dict(device=self.device, dtype=torch.float32)
        return self.run(plan, t.reshape(B).to(**f32).contiguous(), xh.to(**f32).contiguous(),
                        context.to(**f32).contiguous())

    def block_debug(self, plan: BatchPlan, block: int, h: torch.Tensor, x: torch.Tensor, x0: torch.Tensor):
        """One EquivariantBlock on compact arrays (kernel-level parity pin)."""
        L = _lib.lib()
        h = h.to(self.device, torch.float32).contiguous().clone()
        x = x.to(self.device, torch.float32).contiguous().clone()
        x0 = x0.to(self.device, torch.float32).contiguous()
        _lib.check(L.mcg_egnn_block_debug(self.handle, plan.handle, int(block), _lib.dptr(h), _lib.dptr(x),
                                          _lib.dptr(x0), _lib.current_stream_ptr(self.device)), "mcg_egnn_block_debug")
        return h, x

    def gcl_debug(self, plan: BatchPlan, layer: int, h: torch.Tensor, x: torch.Tensor, x0: torch.Tensor):
        """One GCL layer (egnn.py:70-85; layer = 2 * block + {0, 1}) on compact arrays, returning the plan's internal
        buffers: h_out [M,420], pab [M,864] (first edge layer per node: Wa h + b1 | Wb h), agg [M,420],
        hidden [M,420] (SiLU of the node MLP's first layer) - kernel-level parity pins."""
        L = _lib.lib()
        f32 = dict(device=self.device, dtype=torch.float32)
        M = plan.n_real_nodes
        h, x, x0 = h.to(**f32).contiguous(), x.to(**f32).contiguous(), x0.to(**f32).contiguous()
        st = _lib.current_stream_ptr(self.device)
        _lib.check(L.mcg_egnn_gcl_debug(self.handle, plan.handle, int(layer), _lib.dptr(h), _lib.dptr(x), _lib.dptr(x0), st),
                   "mcg_egnn_gcl_debug")
        out = {}
        for name, which, width in (("h_out", 0, 432), ("pab", 1, 864), ("agg", 2, 432), ("hidden", 3, 432)):
            buf = torch.empty(M, width, **f32)
            _lib.check(L.mcg_plan_peek(plan.handle, which, _lib.dptr(buf), st), "mcg_plan_peek")
            out[name] = buf
        out["h_out"], out["agg"], out["hidden"] = out["h_out"][:, :420], out["agg"][:, :420], out["hidden"][:, :420]
        return out

    def __del__(self):
        try:
            self._plans.clear()
            if self._h:
                _lib.lib().mcg_egnn_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:  # noqa: BLE001
            pass
