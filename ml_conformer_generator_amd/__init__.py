"""MI355X-native denoising hot path of mlconfgen (EGNN sampler + AdjMatSeer GCN).

Drop-in for `mlconfgen.MLConformerGenerator` on that path; all per-step
arithmetic runs in hand-written HIP kernels (gfx950) behind the C-ABI declared in
`include/mlconfgen_hip.h`.  There is no CPU fallback: constructing the generator
without the built library raises.
"""
from .config import (ATOM_DECODER, CONTEXT_NORMS, DIMENSION, MAX_N_NODES,
                     MIN_N_NODES, NUM_BOND_TYPES)

__all__ = ["MLConformerGenerator", "ATOM_DECODER", "CONTEXT_NORMS", "DIMENSION",
           "MAX_N_NODES", "MIN_N_NODES", "NUM_BOND_TYPES"]


def __getattr__(name):
    if name == "MLConformerGenerator":
        from .conformer_generator import MLConformerGenerator
        return MLConformerGenerator
    raise AttributeError(name)
