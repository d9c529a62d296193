"""MI355X-native denoising hot path of mlconfgen (EGNN sampler + AdjMatSeer GCN).

Drop-in for `mlconfgen.MLConformerGenerator` on that path; all per-step
arithmetic runs in hand-written HIP kernels (gfx950) behind the C-ABI declared in
`include/mlconfgen_hip.h`.  There is no CPU fallback: constructing the generator
without the built library raises.
"""
from .config import (ATOM_DECODER, CONTEXT_NORMS, DIMENSION, MAX_N_NODES,
                     MIN_N_NODES, NUM_BOND_TYPES)

__all__ = ["MLConformerGenerator", "evaluate_samples", "ATOM_DECODER", "CONTEXT_NORMS", "DIMENSION",
           "MAX_N_NODES", "MIN_N_NODES", "NUM_BOND_TYPES"]


def __getattr__(name):
    # the reference exports `MLConformerGenerator` and `evaluate_samples` from the package root (mlconfgen/__init__.py)
    if name == "MLConformerGenerator":
        from .conformer_generator import MLConformerGenerator
        return MLConformerGenerator
    if name == "evaluate_samples":
        from .cheminformatics import evaluate_samples
        return evaluate_samples
    raise AttributeError(name)
