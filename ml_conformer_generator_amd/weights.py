"""Checkpoint layout of the two networks on the hot path + synthetic weights.

The trained checkpoints (`edm_moi_chembl_15_39.pt`, `adj_mat_seer_chembl_15_39.pt`)
are not redistributable and are absent offline.  This module
  * declares the reference's state-dict key/shape layout (what
    `conformer_generator.py:90-102` loads), and
  * generates deterministic synthetic weights in exactly that layout, so that
    the same tensors can be fed to the reference (fixture generation), to the
    CPU oracle and to the HIP path.

Synthetic weights are a function of (seed, key) only - they do not depend on
module construction order, so no reference code is needed to rebuild them.
"""
from __future__ import annotations

import math
import zlib
from typing import Dict, List, Tuple

import torch

from .config import (DIMENSION, EGNN_HIDDEN, EGNN_IN_NODE_NF, EGNN_N_BLOCKS,
                     GCN_EMBED, GCN_HIDDEN, GCN_NUM_EMBEDDINGS, NUM_BOND_TYPES)

Spec = List[Tuple[str, Tuple[int, ...], int, str]]  # (key, shape, fan_in, kind)


def edm_spec(hidden: int = EGNN_HIDDEN, in_nf: int = EGNN_IN_NODE_NF,
             n_blocks: int = EGNN_N_BLOCKS) -> Spec:
    """Keys of `EquivariantDiffusion.state_dict()` minus `gamma.gamma`
    (egnn.py:23-36,100-108,239-303)."""
    H = hidden
    s: Spec = []
    p = "dynamics.egnn."
    s.append((p + "embedding.weight", (H, in_nf), in_nf, "w"))
    s.append((p + "embedding.bias", (H,), in_nf, "b"))
    s.append((p + "embedding_out.weight", (in_nf, H), H, "w"))
    s.append((p + "embedding_out.bias", (in_nf,), H, "b"))
    for k in range(n_blocks):
        for g in ("gcl_0", "gcl_1"):
            q = f"{p}e_block_{k}.{g}."
            s.append((q + "edge_mlp.0.weight", (H, 2 * H + 2), 2 * H + 2, "w"))
            s.append((q + "edge_mlp.0.bias", (H,), 2 * H + 2, "b"))
            s.append((q + "edge_mlp.2.weight", (H, H), H, "w"))
            s.append((q + "edge_mlp.2.bias", (H,), H, "b"))
            s.append((q + "node_mlp.0.weight", (H, 2 * H), 2 * H, "w"))
            s.append((q + "node_mlp.0.bias", (H,), 2 * H, "b"))
            s.append((q + "node_mlp.2.weight", (H, H), H, "w"))
            s.append((q + "node_mlp.2.bias", (H,), H, "b"))
            s.append((q + "att_mlp.0.weight", (1, H), H, "w"))
            s.append((q + "att_mlp.0.bias", (1,), H, "b"))
        q = f"{p}e_block_{k}.gcl_equiv."
        s.append((q + "coord_mlp.0.weight", (H, 2 * H + 2), 2 * H + 2, "w"))
        s.append((q + "coord_mlp.0.bias", (H,), 2 * H + 2, "b"))
        s.append((q + "coord_mlp.2.weight", (H, H), H, "w"))
        s.append((q + "coord_mlp.2.bias", (H,), H, "b"))
        s.append((q + "coord_mlp.4.weight", (1, H), H, "coord_out"))
    return s


def adj_mat_seer_spec(dimension: int = DIMENSION, hidden: int = GCN_HIDDEN,
                      embed: int = GCN_EMBED, n_emb: int = GCN_NUM_EMBEDDINGS,
                      n_bond: int = NUM_BOND_TYPES) -> Spec:
    """Keys of `AdjMatSeer.state_dict()` (adj_mat_seer.py:84-102)."""
    s: Spec = []
    for name, fin in (("gcn1", embed), ("gcn2", hidden), ("gcn3", hidden), ("gcn4", hidden)):
        s.append((f"{name}.linear.weight", (hidden, fin), fin, "w"))
        s.append((f"{name}.linear.bias", (hidden,), fin, "b"))
    s.append(("resize.weight", (dimension * n_bond, hidden), hidden, "w"))
    s.append(("resize.bias", (dimension * n_bond,), hidden, "b"))
    s.append(("nodes_embedding.weight", (n_emb, embed), 1, "emb"))
    s.append(("nodes_coord_fc.weight", (dimension * embed, dimension), dimension, "w"))
    s.append(("nodes_coord_fc.bias", (dimension * embed,), dimension, "b"))
    for name, fin in (("gcn1_dm", embed), ("gcn2_dm", hidden), ("gcn3_dm", hidden)):
        s.append((f"{name}.linear.weight", (hidden, fin), fin, "w"))
        s.append((f"{name}.linear.bias", (hidden,), fin, "b"))
    s.append(("dm_resize.weight", (1, hidden), hidden, "w"))
    s.append(("dm_resize.bias", (1,), hidden, "b"))
    s.append(("dm_nodes_embedding.weight", (n_emb, embed), 1, "emb"))
    return s


def _key_seed(seed: int, key: str) -> int:
    return (seed * 1000003 + zlib.crc32(key.encode())) % (2**31 - 1)


def synth_state_dict(spec: Spec, seed: int, coord_out_gain: float = 0.05,
                     weight_gain: float = 1.0) -> Dict[str, torch.Tensor]:
    """Deterministic fp32 weights: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for Linear
    weights/biases (the nn.Linear family of scales), N(0,1) for embeddings, and
    U(+-coord_out_gain) for the coordinate head - large enough that the
    coordinate branch is exercised (the reference initialises it with gain 0.001,
    egnn.py:100-101, which would hide errors there)."""
    out: Dict[str, torch.Tensor] = {}
    for key, shape, fan_in, kind in spec:
        g = torch.Generator(device="cpu")
        g.manual_seed(_key_seed(seed, key))
        if kind == "emb":
            t = torch.randn(shape, generator=g, dtype=torch.float32)
        elif kind == "coord_out":
            t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * coord_out_gain
        else:
            bound = weight_gain / math.sqrt(fan_in)
            t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * bound
        out[key] = t.contiguous()
    return out


# Per-tensor gains of the default synthetic EGNN weights ("v2").  Chosen with tools/parity_sensitivity.py so that
# every golden fixture can SEE the arithmetic it pins: Linear layers variance-preserving (gain sqrt(3) ~ 1.7 on the
# nn.Linear U(+-1/sqrt(fan_in)) family, instead of a contractive 0.3 under which messages were ~1e-2 of h), attention
# weights large enough that the gate actually varies over edges, the aggregated-message half of the node MLP's first
# layer boosted against the /100 normalisation, and a small output layer / coordinate head so that an UNTRAINED
# denoiser does not blow the ancestral sampler up (|z| still reaches ~1e3 through the 1/alpha_ts growth, finite in
# fp32 in the reference itself).
RECIPE_V2 = dict(linear=1.7, att=4.0, att_bias=10.0, agg=4.0, node_out=0.5, embed_out=0.3, coord_out=0.05, dist=1.0)
# "v2d": the same with the two squared-distance columns of every edge / coordinate MLP damped.  Fixtures whose
# trajectory passes through |x| ~ 1e3 (resampling and inpainting repeat the 1/alpha_ts amplification of the first
# step) need it: the untrained coordinate head otherwise feeds x -> d^2 ~ x^2 -> x back on itself and overflows fp32
# (in the reference too; a trained head keeps phi small).
RECIPES = {"v2": RECIPE_V2, "v2d": dict(RECIPE_V2, dist=0.02)}


def synth_edm_state_dict(seed: int = 1234, weight_gain: float = None, recipe: str = "v2", **kw) -> Dict[str, torch.Tensor]:
    """Deterministic synthetic EGNN weights in the reference's checkpoint layout.
    Default: the "v2" recipe above.  `weight_gain=g` (legacy form): one uniform gain on every Linear tensor."""
    if weight_gain is not None:
        sd = synth_state_dict(edm_spec(), seed, weight_gain=weight_gain, **kw)
    else:
        if recipe not in RECIPES:
            raise ValueError(f"unknown synthetic weight recipe {recipe!r}")
        r = RECIPES[recipe]
        sd = synth_state_dict(edm_spec(), seed, weight_gain=1.0, coord_out_gain=kw.pop("coord_out_gain", r["coord_out"]), **kw)
        H = EGNN_HIDDEN
        for k, v in sd.items():
            if k.endswith("att_mlp.0.weight"):
                v *= r["att"]
            elif k.endswith("att_mlp.0.bias"):
                v *= r["att_bias"]
            elif k.endswith("node_mlp.0.weight"):
                v *= r["linear"]
                v[:, H:] *= r["agg"]
            elif k.endswith("node_mlp.2.weight"):
                v *= r["node_out"]
            elif k.endswith("embedding_out.weight"):
                v *= r["embed_out"]
            elif k.endswith("coord_mlp.4.weight"):
                pass
            elif k.endswith("edge_mlp.0.weight") or k.endswith("coord_mlp.0.weight"):
                v *= r["linear"]
                v[:, 2 * H:] *= r["dist"]
            elif k.endswith(".weight"):
                v *= r["linear"]
    # The checkpoint also carries the training-time 1000-step schedule
    # (`gamma.gamma`, length 1001); it is replaced right after loading
    # (conformer_generator.py:105-113) so only its presence/shape matters.
    from .schedule import gamma_table
    sd["gamma.gamma"] = gamma_table(1000, 1e-5)
    return sd


def synth_adj_mat_seer_state_dict(seed: int = 4321, **kw) -> Dict[str, torch.Tensor]:
    return synth_state_dict(adj_mat_seer_spec(), seed, **kw)


def check_state_dict(sd: Dict[str, torch.Tensor], spec: Spec, what: str) -> None:
    """Strict key/shape check, mirroring `load_state_dict(strict=True)`."""
    missing = [k for k, *_ in spec if k not in sd]
    if missing:
        raise RuntimeError(f"{what}: missing keys in state_dict: {missing[:5]}{'...' if len(missing) > 5 else ''}")
    for k, shape, *_ in spec:
        if tuple(sd[k].shape) != tuple(shape):
            raise RuntimeError(f"{what}: size mismatch for {k}: {tuple(sd[k].shape)} vs {tuple(shape)}")
