"""CPU oracle for the mlconfgen denoising hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain PyTorch-CPU restatement (fp32 aten ops, same op
sequence and cost as the reference) of:
    EGNNDynamics / EGNN / EquivariantBlock / GCL / EquivariantUpdate   (egnn.py)
    EquivariantDiffusion sampler: forward / inpaint / merge_fragments  (equivariant_diffusion.py)
    AdjMatSeer / GraphConv                                             (adj_mat_seer.py)
    tensor-only input construction                                     (utils/mol_utils.py)
of /root/reference/src/mlconfgen (snapshot 2025-07-04).  Every function cites the
reference file:line it follows.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import it, and only as the checker / timed CPU baseline.  The product package
`ml_conformer_generator_amd` never imports it and has no CPU fallback.

Parity pin: the reference ships no tests or golden vectors, and
its trained weights are unavailable offline (F4).  The oracle is pinned against
outputs of the reference itself, imported in the build container by
`tools/make_golden.py` (package-shell import, RDKit stubbed) with seeded
synthetic weights in the reference's state-dict layout; those outputs are
committed under `tests/golden/` and `tests/test_oracle_golden.py` checks the
oracle against every one of them.  With the real trained weights: parity
unpinned until the weights are supplied out-of-band (re-run the same generator).
The RDKit-owned stages (connectivity guess, canonical order, standardisation) are
outside this oracle: parity unpinned at the RDKit boundary.
"""
