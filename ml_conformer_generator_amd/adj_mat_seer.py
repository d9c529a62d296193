"""AdjMatSeer GCN behind the reference's second operator seam (adj_mat_seer.py:60-165).

`forward(elements, dist_mat, adj_mat) -> [B,42,42,5]` keeps the reference contract
(ONNX names `elements, dist_mat, adj_mat -> out`, onnx_export_utils.py:136-137);
`bond_orders(...)` additionally returns the consumer's argmax (mol_utils.py:210) as
int8 straight from the device.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from .config import (DIMENSION, GCN_EMBED, GCN_HIDDEN, GCN_NUM_EMBEDDINGS, NUM_BOND_TYPES)
from .weights import adj_mat_seer_spec


class AdjMatSeer(torch.nn.Module):
    def __init__(self, dimension: int = DIMENSION, n_hidden: int = GCN_HIDDEN, embedding_dim: int = GCN_EMBED,
                 num_embeddings: int = GCN_NUM_EMBEDDINGS, num_bond_types: int = NUM_BOND_TYPES,
                 device: torch.device = torch.device("cuda:0")):
        super().__init__()
        if (dimension, n_hidden, embedding_dim, num_embeddings, num_bond_types) != (
                DIMENSION, GCN_HIDDEN, GCN_EMBED, GCN_NUM_EMBEDDINGS, NUM_BOND_TYPES):
            raise ValueError("the HIP kernels are specialised for the published AdjMatSeer architecture "
                             "(dimension=42, n_hidden=2048, embedding_dim=64, num_embeddings=36, num_bond_types=5)")
        self.dimension, self.embedding_dim, self.num_bond_types = dimension, embedding_dim, num_bond_types
        self.device = torch.device(device)
        self._h = C.c_void_p()

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True):
        """Loads a reference `AdjMatSeer.state_dict()` (conformer_generator.py:97-102)."""
        L = _lib.lib()
        tensors = []
        for key, shape, _, _ in adj_mat_seer_spec():
            if key not in state_dict:
                raise RuntimeError(f"Missing key(s) in state_dict: \"{key}\"")
            t = state_dict[key].detach().to("cpu", torch.float32).contiguous()
            if tuple(t.shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {key}: {tuple(t.shape)} vs {tuple(shape)}")
            tensors.append(t)
        arr = _lib.host_ptr_array(tensors)
        if self._h:
            L.mcg_gcn_destroy(self._h)
            self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.mcg_gcn_create(arr, len(tensors), C.byref(self._h)), "mcg_gcn_create")
        return None

    def _run(self, elements, dist_mat, adj_mat, want_logits: bool, want_bond: bool
             ) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
        if not self._h:
            raise _lib.McgError("AdjMatSeer has no weights loaded")
        B, d = int(elements.shape[0]), self.dimension
        el = elements.to(self.device, torch.int64).contiguous()
        dm = dist_mat.to(self.device, torch.float32).contiguous()
        am = adj_mat.to(self.device, torch.float32).contiguous()
        if tuple(el.shape) != (B, d) or tuple(dm.shape) != (B, d, d) or tuple(am.shape) != (B, d, d):
            raise ValueError(f"expected elements[B,{d}], dist_mat[B,{d},{d}], adj_mat[B,{d},{d}]")
        logits = torch.empty((B, d, d, self.num_bond_types), device=self.device, dtype=torch.float32) if want_logits else None
        bond = torch.empty((B, d, d), device=self.device, dtype=torch.int8) if want_bond else None
        if B > 0:
            _lib.check(_lib.lib().mcg_gcn_forward(self._h, _lib.dptr(el), _lib.dptr(dm), _lib.dptr(am),
                                                  _lib.dptr(logits), _lib.dptr(bond), B,
                                                  _lib.current_stream_ptr(self.device)), "mcg_gcn_forward")
        return logits, bond

    @torch.no_grad()
    def forward(self, elements: torch.Tensor, dist_mat: torch.Tensor, adj_mat: torch.Tensor) -> torch.Tensor:
        return self._run(elements, dist_mat, adj_mat, True, False)[0]

    @torch.no_grad()
    def bond_orders(self, elements, dist_mat, adj_mat, with_logits: bool = False):
        """int8 [B,42,42] argmax over bond classes (0 none, 1 single, 2 double, 3 triple, 4 aromatic)."""
        logits, bond = self._run(elements, dist_mat, adj_mat, with_logits, True)
        return (bond, logits) if with_logits else bond

    def check_inputs_seen(self) -> None:
        """Raises IndexError if an element id outside [0,36) was ever fed (nn.Embedding semantics)."""
        if self._h and _lib.lib().mcg_gcn_check(self._h) != 0:
            raise IndexError("index out of range in self (element id outside the embedding table)")

    def __del__(self):
        try:
            if self._h:
                _lib.lib().mcg_gcn_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:  # noqa: BLE001
            pass
