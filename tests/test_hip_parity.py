"""GPU parity tests: HIP path (through the C ABI) vs the CPU oracle and the golden vectors
captured from the reference.  Run on the MI355X box with `-m gpu`.

Stated fp32 tolerances: one network call  rtol 1e-4 / atol 1e-5 (relative to
the tensor's scale); sampler trajectories with identical noise rel 1e-3 of max|z|;
atom types and adjacency indices exact.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN as GOLD
from conftest import REPO, TapeNoise, load_golden, sd_for
from parity_tolerance import close as _close
from parity_tolerance import LONG_TRAJ_REL, traj_violation, violation

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def edge_mask_of(node_mask):
    B, N, _ = node_mask.shape
    nm = node_mask.squeeze(2)
    em = nm.unsqueeze(1) * nm.unsqueeze(2) * (1 - torch.eye(N)).unsqueeze(0)
    return em.reshape(B * N * N, 1)


def close(a, b, rtol=1e-4, atol=1e-5):
    """Stated per-call tolerance (tests/parity_tolerance.py): |a - b| <= atol * S + rtol * |b| with S the REAL magnitude
    of the element's channel group; [.., 11] tensors are split into coordinates / velocity (0..2) and features."""
    return _close(a, b, rtol, atol, split=3 if b.shape[-1] == 11 else None)


_DYN = {}


def dyn_for(g, mode="f32"):
    """EGNNDynamics holding the weights a golden fixture was generated with (one instance per operand mode)."""
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    key = (int(g["weight_seed"]), str(g["weight_recipe"]) if "weight_recipe" in g else "v2", mode)
    if key not in _DYN:
        d = EGNNDynamics(device=DEV)
        d.load_reference_state_dict(sd_for(g))
        d.set_precision(mode)
        _DYN[key] = d
    return _DYN[key]


@pytest.fixture(scope="module")
def dyn(edm_sd):
    return dyn_for({"weight_seed": 1234, "weight_recipe": "v2"})


@pytest.fixture(scope="module")
def sampler_factory(dyn):
    from ml_conformer_generator_amd.equivariant_diffusion import EquivariantDiffusion, PredefinedNoiseSchedule

    def make(T, g=None, mode="f32"):
        gm = EquivariantDiffusion(dynamics=dyn if (g is None and mode == "f32") else dyn_for(g if g is not None else {"weight_seed": 1234, "weight_recipe": "v2"}, mode),
                                  in_node_nf=8, timesteps=1000, noise_precision=1e-5)
        gm.gamma = PredefinedNoiseSchedule(timesteps=T, precision=1e-5)
        gm.T = T
        return gm
    return make


def test_library_loaded_is_in_tree():
    import os
    from ml_conformer_generator_amd import _lib
    _lib.lib()
    maps = open("/proc/self/maps").read()
    assert os.path.realpath(_lib.LIB_PATH) in maps


def test_single_block_vs_golden(dyn):
    g = load_golden("block3_b2n20.npz")
    nm = g["node_mask"].squeeze(2)
    n_nodes = nm.sum(1).to(torch.int32)
    plan = dyn.plan(n_nodes, nm.shape[1])
    real = nm.reshape(-1) > 0
    h, x = dyn.block_debug(plan, 3, g["h_in"][real], g["x_in"][real], g["x0"][real])
    ok, err, sc = close(h, g["h_out"][real])
    assert ok, f"h err {err} scale {sc}"
    ok, err, sc = close(x, g["x_out"][real])
    assert ok, f"x err {err} scale {sc}"


def test_gcl_internals_vs_oracle(dyn, edm_sd):
    """Kernel-level pins of ONE GCL layer (egnn.py:38-85) through mcg_egnn_gcl_debug + mcg_plan_peek: the per-node
    first-layer projections, the gated + masked aggregate (/100), the node MLP's hidden layer and the new h, every
    element, against oracle.gcl()'s internals and the reference's own `h_after_gcl0` / `agg_gcl0` (fixture).
    rtol 1e-4, atol 1e-5 of each tensor's real magnitude."""
    import torch.nn.functional as F
    from oracle import egnn_oracle as EO
    g = load_golden("block3_b2n20.npz")
    nm = g["node_mask"]
    B, N, _ = nm.shape
    real = nm.reshape(-1) > 0
    nmf, emf = nm.reshape(B * N, 1), edge_mask_of(nm)
    row, col = EO.dense_edge_index(N, B)
    d0, _ = EO.pair_geometry(g["x0"], row, col)
    d1, _ = EO.pair_geometry(g["x_in"], row, col)
    p = "dynamics.egnn.e_block_3.gcl_0."
    h_ref, m_ref, msg_ref, agg_ref = EO.gcl(edm_sd, p, g["h_in"], row, col, torch.cat([d1, d0], 1), nmf, emf)
    w1, b1 = edm_sd[p + "edge_mlp.0.weight"], edm_sd[p + "edge_mlp.0.bias"]
    pab_ref = torch.cat([F.linear(g["h_in"], w1[:, :420], b1), F.linear(g["h_in"], w1[:, 420:840])], 1)
    hid_ref = F.silu(F.linear(torch.cat([g["h_in"], agg_ref], 1), edm_sd[p + "node_mlp.0.weight"], edm_sd[p + "node_mlp.0.bias"]))
    aggs = {}
    for mode in (-1, 0, 1):                           # quarter-tile units (auto at this size), four-tile units, k_edge_ns
        plan = dyn.plan(nm.sum(1).reshape(-1).to(torch.int32), N, edge_mt=1)
        plan.set_latency_mode(mode)
        got = dyn.gcl_debug(plan, 6, g["h_in"][real], g["x_in"][real], g["x0"][real])
        plan.set_latency_mode(-1)
        aggs[mode] = got["agg"].cpu()
        pab = got["pab"].cpu()
        pab = torch.cat([pab[:, :420], pab[:, 432:852]], 1)
        for name, a, b in (("pab", pab, pab_ref[real]), ("agg", got["agg"], agg_ref[real]),
                           ("agg vs reference", got["agg"], g["agg_gcl0"][real]), ("hidden", got["hidden"], hid_ref[real]),
                           ("h", got["h_out"], h_ref[real]), ("h vs reference", got["h_out"], g["h_after_gcl0"][real])):
            ok, err, sc = close(a, b)
            assert ok, f"mode {mode} {name}: err {err} scale {sc}"
    assert float(agg_ref.abs().max()) > 0.05 and float(msg_ref.abs().max()) > 0.1      # the pins carry signal
    # ORDER of the /100 (egnn.py:435): the reference and k_edge_ns's combine divide the finished sum, the workgroup-level
    # paths divide every partial row (one per unit an atom's rows run through) before the consumer adds them -
    # a/100 + b/100 vs (a + b)/100.  m_ij is bit-identical between the edge paths (columns 416..419 of the four-tile body
    # excepted: four k-slices on v_mfma_f32_4x4x1); agg differs by those roundings only:
    sc_agg = float(agg_ref.abs().max())
    for a in (-1, 0):
        d = float((aggs[a] - aggs[1]).abs().max())
        assert d <= 4e-7 * sc_agg, (a, d, sc_agg)


@pytest.mark.parametrize("sizes", [[2, 1] * 8, [1, 2, 3] * 6 + [2, 2, 1, 1, 3], [2] * 33, [3, 1, 1, 2, 17, 1, 2]])
def test_tiny_molecules_fill_tiles_with_many_segments(dyn, edm_sd, sizes):
    """1-, 2- and 3-atom molecules: a 16-row edge tile then spans up to 16 row-owning atoms (and more node indices,
    1-atom molecules own no rows) - segment ids are ranks among row-owning atoms, never above 15."""
    from oracle import egnn_oracle as EO
    from oracle import host_oracle as HO
    torch.manual_seed(16)
    sz = torch.tensor(sizes)
    B, N = len(sizes), int(max(sizes)) + 1
    nm, em = HO.masks_from_sizes(sz, N)
    z = torch.randn(B, N, 11) * nm
    ctx = torch.randn(B, 1, 3).repeat(1, N, 1) * nm
    t = torch.full((B, 1), 0.4)
    ref = EO.egnn_dynamics(edm_sd, t, z, nm, em, ctx)
    for mode in (-1, 0, 1):
        plan = dyn.plan(sz, N, edge_mt=1)
        plan.set_latency_mode(mode)
        out = dyn.run(plan, t.reshape(-1).to(DEV), z.to(DEV), ctx.to(DEV))
        plan.set_latency_mode(-1)
        ok, err, sc = close(out, ref)
        assert ok, f"mode {mode}: err {err} scale {sc}"
    # wider units cannot hold such batches: refused, not silently wrong
    from ml_conformer_generator_amd import _lib
    if sizes == [2] * 33:                  # 64 rows = 64 row-owning atoms in one unit
        with pytest.raises(_lib.McgError):
            dyn.plan(sz, N, edge_mt=4)


@pytest.mark.parametrize("tag", ["b2n20", "b4n19", "b3n39", "b3n27_x30"])
def test_dynamics_seam_vs_golden(dyn, tag):
    g = load_golden(f"dynamics_{tag}.npz")
    nm = g["node_mask"]
    out = dyn(g["t"].to(DEV), g["xh"].to(DEV), nm.to(DEV), edge_mask_of(nm).to(DEV), g["context"].to(DEV))
    ok, err, sc = close(out, g["out"])
    assert ok, f"err {err} scale {sc}"
    # padded slots are exactly zero, inputs untouched
    assert float((out.cpu() * (1 - nm)).abs().max()) == 0.0


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("tag", ["b2n20", "b3n39", "b3n27_x30"])
def test_both_edge_kernels_on_the_golden_shapes(dyn, tag, mode):
    """Four-tile workgroups only (mode 0) and the stand-alone column-split kernel (mode 1) on the same small golden
    inputs (auto mode runs these sizes on the quarter-tile units of the throughput kernel: test_dynamics_seam_vs_golden)."""
    g = load_golden(f"dynamics_{tag}.npz")
    nm = g["node_mask"]
    B, N, _ = nm.shape
    plan = dyn.plan(nm.sum(1).reshape(-1).to(torch.int32), N, edge_mt=1)
    plan.set_latency_mode(mode)
    out = dyn.run(plan, g["t"].reshape(B).to(DEV), g["xh"].to(DEV), g["context"].to(DEV))
    plan.set_latency_mode(-1)
    ok, err, sc = close(out, g["out"])
    assert ok, f"mode {mode}: err {err} scale {sc}"


@pytest.mark.parametrize("four_tile", [-1, "all", 16, 40])
def test_mixed_four_tile_and_quarter_tile_units_vs_oracle(dyn, edm_sd, four_tile):
    """One edge launch whose first workgroups take four tiles and whose last take one tile each
    (`mcg_plan_opts.four_tile_units`): ragged molecules, so atoms straddle two four-tile units, a four-tile and a quarter-tile unit, or
    three quarter-tile units (the node GEMM's three-row gather)."""
    from oracle import egnn_oracle as EO
    from oracle import host_oracle as HO
    torch.manual_seed(23)
    sz = torch.randint(12, 34, (10,))            # n - 1 <= 32 rows: an atom spans at most three 16-row units
    sz[3] = 33
    B, N = sz.numel(), 40
    nm, em = HO.masks_from_sizes(sz, N)
    z = torch.randn(B, N, 11) * nm
    ctx = torch.randn(B, 1, 3).repeat(1, N, 1) * nm
    t = torch.full((B, 1), 0.6)
    ref = EO.egnn_dynamics(edm_sd, t, z, nm, em, ctx)
    from ml_conformer_generator_amd import _lib
    plan = dyn.plan(sz, N, edge_mt=1, four_tile_units=_lib.ALL_FOUR_TILE if four_tile == "all" else four_tile)
    out = dyn.run(plan, t.reshape(-1).to(DEV), z.to(DEV), ctx.to(DEV))
    ok, err, sc = close(out, ref)
    assert ok, f"four_tile_units={four_tile}: err {err} scale {sc}"
    assert float((out.cpu() * (1 - nm)).abs().max()) == 0.0


def test_largest_molecules_span_four_quarter_tile_units(dyn, edm_sd):
    """42-atom molecules (the reference's pad width): an atom's 41 edge rows run through up to four one-tile units, so
    the node GEMM gathers four rows of the workgroup-level sums and the coordinate update adds four."""
    from oracle import egnn_oracle as EO
    from oracle import host_oracle as HO
    torch.manual_seed(29)
    sz = torch.tensor([42, 41, 37, 42, 35])
    B, N = sz.numel(), 42
    nm, em = HO.masks_from_sizes(sz, N)
    z = torch.randn(B, N, 11) * nm
    ctx = torch.randn(B, 1, 3).repeat(1, N, 1) * nm
    t = torch.full((B, 1), 0.3)
    ref = EO.egnn_dynamics(edm_sd, t, z, nm, em, ctx)
    for four_tile in (-1, 24):
        plan = dyn.plan(sz, N, edge_mt=1, four_tile_units=four_tile)
        out = dyn.run(plan, t.reshape(-1).to(DEV), z.to(DEV), ctx.to(DEV))
        ok, err, sc = close(out, ref)
        assert ok, f"four_tile_units={four_tile}: err {err} scale {sc}"


@pytest.mark.parametrize("n_ranges", [1, 2, 3])
def test_dynamics_molecule_ranges_agree_with_oracle(dyn, edm_sd, n_ranges):
    """The same ragged batch as one plan and cut into 2 / 3 molecule ranges on separate HIP streams
    (`mcg_plan_opts.n_ranges`; the library itself splits from 3 600 edge tiles on) against the oracle."""
    from oracle import egnn_oracle as EO
    from oracle import host_oracle as HO
    torch.manual_seed(5)
    sizes = torch.tensor([15, 39, 22, 16, 31, 27] * 2)
    B, N = sizes.numel(), 39
    nm, em = HO.masks_from_sizes(sizes, N)
    z = torch.randn(B, N, 11) * nm * 3.0
    ctx = torch.randn(1, 1, 3).repeat(B, N, 1) * nm
    t = torch.full((B, 1), 0.37)
    ref = EO.egnn_dynamics(edm_sd, t, z, nm, em, ctx)
    plan = dyn.plan(sizes, N, edge_mt=1, n_ranges=n_ranges)
    assert plan.edge_mt == 1 and plan.n_ranges == n_ranges
    out = dyn.run(plan, t.reshape(-1).to(DEV), z.to(DEV), ctx.to(DEV))
    ok, err, sc = close(out, ref)
    assert ok, f"err {err} scale {sc}"
    out2 = dyn.run(plan, t.reshape(-1).to(DEV), z.to(DEV), ctx.to(DEV))
    assert torch.equal(out, out2)


def test_model_options_and_measurement_hooks(edm_sd):
    """`mcg_egnn_set_option` (node-GEMM tile widths, split-operand GEMMs on / off) changes launch shapes, never results
    beyond fp32 re-association: every setting against the oracle under the stated tolerance.  `mcg_bench_edge_incall`
    times 18 + 9 edge launches per call and range and leaves the plan usable."""
    import numpy as np
    from ml_conformer_generator_amd import _lib
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    from oracle import egnn_oracle as EO
    from oracle import host_oracle as HO
    torch.manual_seed(17)
    sizes = torch.tensor([21, 27, 15, 33, 18])
    B, N = sizes.numel(), 33
    nm, em = HO.masks_from_sizes(sizes, N)
    z = torch.randn(B, N, 11) * nm
    ctx = torch.randn(B, 1, 3).repeat(1, N, 1) * nm
    t = torch.full((B, 1), 0.45)
    ref = EO.egnn_dynamics(edm_sd, t, z, nm, em, ctx)
    d = EGNNDynamics(device=DEV)
    d.load_reference_state_dict(edm_sd)
    L = _lib.lib()

    def launches(reset=True):
        c = np.zeros(32, dtype=np.int64)
        _lib.check(L.mcg_debug_gemm_launches(c.ctypes.data, 1 if reset else 0), "mcg_debug_gemm_launches")
        return c.reshape(4, 8)

    # fp32 at 114 atoms: every node GEMM takes the 16-row-tile kernel, which MCG_OPT_GEMM_RN does not steer (round-3
    # advisor: the old loop over this option in fp32 mode tested nothing) - said here, with the launch counters
    launches()
    ok, err, sc = close(d(t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV)), ref)
    c = launches()
    assert ok and int(c[2].sum()) == 63 and int(c[0].sum()) == 0, (err, sc, c)
    # bf16 mode: the 32-row kernel runs all 63 GEMMs and the option picks its wave tile width.  The option is set through
    # the C ABI AFTER the plan captured its graph, without dropping the plan: the option epoch in the graph key must
    # force a re-capture (round-3 advisor), and the counters must show the new width.
    d.set_precision("bf16")
    pl = d.plan(sizes, N)
    zt0, ct0, tt0 = z.to(DEV), ctx.to(DEV), t.reshape(-1).to(DEV).contiguous()
    outs = {}
    for rn in (0, 1, 2, 3):
        _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_GEMM_RN, rn), "mcg_egnn_set_option")      # plan and graph stay
        launches()
        outs[rn] = d.run(pl, tt0, zt0, ct0).clone()
        c = launches()
        assert int(c[1].sum()) == 63, (rn, c)                      # re-captured: 63 launches issued again
        if rn:
            assert int(c[1][rn]) == 63, (rn, c)                    # ... all of them at the requested width
        launches()
        d.run(pl, tt0, zt0, ct0)
        assert int(launches().sum()) == 0                          # unchanged options: the graph is replayed
        assert float((outs[rn] - ref.to(DEV)).abs().max()) <= 3e-2 * float(ref.abs().max())
    for rn in (1, 2, 3):                                           # same k order per output element at every width
        assert torch.equal(outs[rn], outs[1])
    _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_GEMM_RN, 0), "mcg_egnn_set_option")
    d.set_precision("f32")
    with pytest.raises(_lib.McgError):
        d.set_option(_lib.OPT_GEMM_RN, 7)
    with pytest.raises(_lib.McgError):
        d.set_option(99, 1)
    d.set_precision("f32x6")
    for x6_gemm, rn in ((0, 0), (1, 1), (1, 3), (1, 0)):
        d.set_option(_lib.OPT_X6_GEMM, x6_gemm)
        d.set_option(_lib.OPT_GEMM_X6_RN, rn)
        ok, err, sc = close(d(t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV)), ref)
        assert ok, (x6_gemm, rn, err, sc)
    d.set_precision("f32")
    plan = d.plan(sizes, N)
    us = np.zeros(4, dtype=np.float32)
    zt, ct, tt = z.to(DEV), ctx.to(DEV), t.reshape(-1).to(DEV)
    out = torch.empty_like(zt)
    _lib.check(_lib.lib().mcg_bench_edge_incall(d.handle, plan.handle, _lib.dptr(tt), _lib.dptr(zt), _lib.dptr(ct), _lib.dptr(out), 2,
                                                us.ctypes.data, _lib.current_stream_ptr(DEV)), "mcg_bench_edge_incall")
    assert int(us[2]) == 2 * 18 * plan.n_ranges and int(us[3]) == 2 * 9 * plan.n_ranges
    assert 1.0 < float(us[0]) < 1000.0 and 1.0 < float(us[1]) < 1000.0          # microseconds per edge launch
    ok, err, sc = close(out, ref)                                                 # the timed calls computed the real thing
    assert ok, (err, sc)
    ok, err, sc = close(d.run(plan, tt, zt, ct), ref)                             # and the plan still replays its graph
    assert ok, (err, sc)


def test_dynamics_full_gain_weights_vs_oracle():
    """Unit-gain synthetic weights (messages / aggregates are O(1), so every operand of the node
    MLP carries weight in the result) on a ragged batch, against the oracle."""
    from ml_conformer_generator_amd import weights as W
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    from oracle import egnn_oracle as EO
    from oracle import host_oracle as HO
    sd = W.synth_edm_state_dict(99, weight_gain=1.0, coord_out_gain=0.2)
    d = EGNNDynamics(device=DEV)
    d.load_reference_state_dict(sd)
    torch.manual_seed(8)
    sizes = torch.tensor([17, 33, 15, 26])
    N = 33
    nm, em = HO.masks_from_sizes(sizes, N)
    z = torch.randn(4, N, 11) * nm
    ctx = torch.randn(4, 1, 3).repeat(1, N, 1) * nm
    t = torch.full((4, 1), 0.25)
    ref = EO.egnn_dynamics(sd, t, z, nm, em, ctx)
    out = d(t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV))
    ok, err, sc = close(out, ref)
    assert ok, f"err {err} scale {sc}"
    assert sc > 0.5


def test_small_fragments_and_degenerate_sizes(dyn, edm_sd):
    """n = 1 (no edges), n = 2, n = 7 (fragment generation sizes) next to a large molecule."""
    from oracle import egnn_oracle as EO
    from oracle import host_oracle as HO
    torch.manual_seed(6)
    sizes = torch.tensor([1, 2, 7, 30, 3])
    N = 31
    nm, em = HO.masks_from_sizes(sizes, N)
    z = torch.randn(5, N, 11) * nm
    ctx = torch.randn(5, 1, 3).repeat(1, N, 1) * nm
    t = torch.full((5, 1), 0.9)
    ref = EO.egnn_dynamics(edm_sd, t, z, nm, em, ctx)
    out = dyn(t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV))
    ok, err, sc = close(out, ref)
    assert ok, f"err {err} scale {sc}"
    # the opt-in operand modes keep working on such batches (their 64-row kernel needs >= 6 atoms per molecule:
    # bf16 falls back to its 16-row kernel, f32x6 to the exact fp32 kernels)
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    d2 = EGNNDynamics(device=DEV)
    d2.load_reference_state_dict(edm_sd)
    d2.set_precision("f32x6")
    ok, err, sc = close(d2(t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV)), ref)
    assert ok, f"f32x6 err {err} scale {sc}"
    d2.set_precision("bf16")
    o16 = d2(t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV)).cpu()
    assert float((o16 - ref).abs().max()) / float(ref.abs().max()) < 3e-2


def test_non_prefix_mask_rejected(dyn):
    nm = torch.zeros(1, 5, 1)
    nm[0, 1:4] = 1
    with pytest.raises(ValueError):
        dyn(torch.zeros(1, 1), torch.zeros(1, 5, 11), nm, torch.zeros(25, 1), torch.zeros(1, 5, 3))


def test_non_canonical_edge_mask_is_refused_not_ignored(dyn, sampler_factory):
    """The reference multiplies by whatever `edge_mask` it is given (egnn.py:477-478,51,127); the HIP path implements the
    canonical mask of the prefix node mask only, so any other mask must raise - through op seam 1 and through the sampler
    entry points - and the canonical mask (any shape with B*N*N entries), or None, must pass.  The verdict is cached per
    tensor object + version: an in-place edit of an accepted mask is noticed."""
    g = load_golden("dynamics_b4n19.npz")
    nm, xh, ctx, t = g["node_mask"], g["xh"], g["context"], g["t"]
    em = edge_mask_of(nm)
    ref = dyn(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV))
    assert torch.equal(ref, dyn(t.to(DEV), xh.to(DEV), nm.to(DEV), None, ctx.to(DEV)))
    assert torch.equal(ref, dyn(t.to(DEV), xh.to(DEV), nm.to(DEV), em.reshape(nm.shape[0], -1).to(DEV), ctx.to(DEV)))
    bad = em.clone()
    bad[int(torch.nonzero(bad)[5, 0])] = 0.0                                  # one extra zero
    with pytest.raises(ValueError, match="canonical"):
        dyn(t.to(DEV), xh.to(DEV), nm.to(DEV), bad.to(DEV), ctx.to(DEV))
    with pytest.raises(ValueError, match="entries"):
        dyn(t.to(DEV), xh.to(DEV), nm.to(DEV), em[:-1].to(DEV), ctx.to(DEV))
    live = em.to(DEV)
    dyn(t.to(DEV), xh.to(DEV), nm.to(DEV), live, ctx.to(DEV))                  # accepted and cached ...
    live[int(torch.nonzero(live)[0, 0])] = 0.0                                 # ... then edited in place
    with pytest.raises(ValueError, match="canonical"):
        dyn(t.to(DEV), xh.to(DEV), nm.to(DEV), live, ctx.to(DEV))
    gm = sampler_factory(4)
    with pytest.raises(ValueError, match="canonical"):
        gm(nm.to(DEV), bad.to(DEV), ctx.to(DEV), 0)
    x, h = gm(nm.to(DEV), em.to(DEV), ctx.to(DEV), 0)
    assert bool(torch.isfinite(x).all())


def test_mask_verdicts_are_cached_so_a_sampler_loop_pays_no_sync_per_call(dyn, monkeypatch):
    """Round-4 review: `EGNNDynamics.forward` derived the sizes with a device compare + `.cpu()` on EVERY call - one host
    sync per denoiser step for anyone who binds op seam 1 inside the reference's own sampler loop.  Now: once per mask
    tensor (identity + version), never for masks built by this package's `prepare_masks` (tagged), and an in-place edit of
    a checked or tagged mask is noticed."""
    from ml_conformer_generator_amd import egnn as E
    from ml_conformer_generator_amd.mol_utils import prepare_masks
    g = load_golden("dynamics_b4n19.npz")
    nm, xh, ctx, t = g["node_mask"].to(DEV), g["xh"].to(DEV), g["context"].to(DEV), g["t"].to(DEV)
    em = edge_mask_of(g["node_mask"]).to(DEV)
    calls = []
    real = E.sizes_from_node_mask
    monkeypatch.setattr(E, "sizes_from_node_mask", lambda m: (calls.append(1), real(m))[1])
    ref = dyn(t, xh, nm, em, ctx)
    for _ in range(5):                                           # the reference's loop: the same mask tensors every step
        assert torch.equal(dyn(t, xh, nm, em, ctx), ref)
    assert len(calls) == 1
    sizes = g["node_mask"].sum(1).reshape(-1).long()
    nm2, em2 = prepare_masks(sizes, nm.shape[1], DEV)            # the package's own masks: no check at all
    plan = dyn.plan(dyn.sizes_for(nm2), nm.shape[1])
    monkeypatch.setattr(plan, "edge_mask", lambda: (_ for _ in ()).throw(AssertionError("tagged masks need no device compare")))
    assert torch.equal(dyn(t, xh, nm2, em2, ctx), ref) and len(calls) == 1
    monkeypatch.undo()
    em2[int(torch.nonzero(em2)[3, 0])] = 0.0                     # a tagged mask edited in place loses its tag
    with pytest.raises(ValueError, match="canonical"):
        dyn(t, xh, nm2, em2, ctx)
    nm2[0, 0] = 0.0                                              # ... and so does a tagged node mask (no longer a prefix)
    with pytest.raises(ValueError, match="prefix"):
        dyn(t, xh, nm2, None, ctx)
    stats = dyn.release_cached_memory()
    assert stats["cached_bytes"] == 0 and len(dyn._plans) == 0
    assert torch.equal(dyn(t, xh, nm, em, ctx), ref)             # everything is rebuilt on demand


# (mode "f32x6": the opt-in split-operand kernels under the SAME per-step tolerance as the exact path - every golden
#  trajectory of the reference: plain, resampling, inpainting, fragment merge)
@pytest.mark.parametrize("mode", ["f32", "f32x6"])
@pytest.mark.parametrize("name,rs", [("sampler_T20_b4n19.npz", 0), ("sampler_T8_rs1.npz", 1)])
def test_sampler_trajectory_vs_golden(sampler_factory, name, rs, mode):
    g = load_golden(name)
    nm = g["node_mask"]
    gm = sampler_factory(int(g["T"]), g, mode)
    gm.noise_fn = TapeNoise(g["noise"], DEV)
    gm.trace = []
    x, h = gm(nm.to(DEV), edge_mask_of(nm).to(DEV), g["context"].to(DEV), rs)
    assert gm.noise_fn.pos == g["noise"].numel()          # same number / order of draws as the reference
    zt = torch.stack(gm.trace).cpu()
    v = traj_violation(zt, g["z_trace"])                   # 1e-3 of every step's own coordinate / feature magnitude
    assert v <= 1.0, f"trajectory at {v} x tolerance"
    assert traj_violation(x.cpu().unsqueeze(0), g["x"].unsqueeze(0), split=None) <= 1.0
    assert torch.equal(h.cpu().to(torch.int64), g["h"].to(torch.int64))      # atom types exact
    gm.noise_fn, gm.trace = None, None


def test_sampler_step_teacher_forced(sampler_factory):
    """Feed the REFERENCE's z_t into one HIP step and compare z_s: no error compounding."""
    g = load_golden("sampler_T20_b4n19.npz")
    nm = g["node_mask"]
    B, N, _ = nm.shape
    gm = sampler_factory(20)
    per = B * N * 11
    noise = g["noise"]
    run = gm._Run(gm, nm.to(DEV), g["context"])
    worst = 0.0
    for step_idx in (1, 7, 19):            # step_idx-th network call (0-based); z_trace[k] is its output
        s_int = 19 - step_idx
        z_in = g["z_trace"][step_idx - 1].to(DEV).contiguous().clone()
        gm.noise_fn = TapeNoise(noise[(step_idx + 1) * per:(step_idx + 2) * per], DEV)
        z_out = gm._step(run, z_in, s_int).cpu()
        ref = g["z_trace"][step_idx]
        ok, err, sc = close(z_out, ref, rtol=1e-4, atol=1e-5)
        worst = max(worst, err / sc)
        assert ok, f"step {step_idx}: err {err} scale {sc}"
    gm.noise_fn = None


@pytest.mark.parametrize("mode", ["f32", "f32x6"])
def test_inpaint_vs_golden(sampler_factory, mode):
    g = load_golden("inpaint_T5.npz")
    nm = g["node_mask"]
    gm = sampler_factory(int(g["T"]), g, mode)
    gm.noise_fn = TapeNoise(g["noise"], DEV)
    gm.trace = []
    x, h = gm.inpaint(nm.to(DEV), edge_mask_of(nm).to(DEV), g["context"].to(DEV), g["z_known"], g["fixed_mask"], 1, 3)
    assert gm.noise_fn.pos == g["noise"].numel()
    v = traj_violation(torch.stack(gm.trace).cpu(), g["z_trace"])
    assert v <= 1.0, f"trajectory at {v} x tolerance"
    assert traj_violation(x.cpu().unsqueeze(0), g["x"].unsqueeze(0), split=None) <= 1.0
    assert torch.equal(h.cpu().to(torch.int64), g["h"].to(torch.int64))
    gm.noise_fn, gm.trace = None, None


def test_sampler_at_the_judged_step_counts_vs_reference_golden(sampler_factory):
    """Round 5: every other golden trajectory is T <= 20; the bench times T = 100 (configs[1], 101 denoiser calls) and
    configs[4] runs T = 250 with inpainting and resample_steps = 1 (501 calls).  The HIP sampler under the reference's
    recorded tapes at those lengths (`tools/make_golden.py` section 10: contractive weights, a one-ulp change of the
    context moves the reference's own final x by 3e-7 / 6e-7 of max|x|): same number and order of noise draws, every
    recorded latent (each 10th / 50th step) and the final x within the stated trajectory tolerance (1e-3 of the step's own
    channel-group magnitude), atom types exact.
    Round 6: held to LONG_TRAJ_REL = 4e-6 instead of the 1e-3 of the short full-gain trajectories - the mutation check of these
    two contractive fixtures (profiles/round6_parity_sensitivity.txt) needs it, and the fp32 kernels follow them to 7e-7."""
    g = load_golden("e2e_T100_b2n27.npz")
    nm = g["node_mask"]
    gm = sampler_factory(int(g["T"]), g, "f32")
    gm.noise_fn = TapeNoise(g["noise"], DEV)
    gm.trace = []
    x, h = gm(nm.to(DEV), edge_mask_of(nm).to(DEV), g["context"].to(DEV), 0)
    assert gm.noise_fn.pos == g["noise"].numel() and len(gm.trace) == 100
    v = traj_violation(torch.stack(gm.trace).cpu()[g["z_trace_index"].long()], g["z_trace"], rel=LONG_TRAJ_REL)
    assert v <= 1.0, f"T = 100 trajectory at {v} x tolerance"
    vx = traj_violation(x.cpu().unsqueeze(0), g["x"].unsqueeze(0), rel=LONG_TRAJ_REL, split=None)
    assert vx <= 1.0 and torch.equal(h.cpu().to(torch.int64), g["h"].to(torch.int64))
    print(f"T=100: worst recorded latent at {v:.3f} x tolerance, final x at {vx:.3f} (reference amplification of one ulp: "
          f"{float(g['one_ulp_context_rel_dev']):.1e})")
    gm.noise_fn, gm.trace = None, None
    g = load_golden("inpaint_T250_rs1_b2.npz")
    nm = g["node_mask"]
    gm = sampler_factory(int(g["T"]), g, "f32")
    gm.noise_fn = TapeNoise(g["noise"], DEV)
    gm.trace = []
    x, h = gm.inpaint(nm.to(DEV), edge_mask_of(nm).to(DEV), g["context"].to(DEV), g["z_known"], g["fixed_mask"], 1, 3)
    assert gm.noise_fn.pos == g["noise"].numel() and len(gm.trace) == int(g["n_sampler_steps"])
    v = traj_violation(torch.stack(gm.trace).cpu()[g["z_trace_index"].long()], g["z_trace"], rel=LONG_TRAJ_REL)
    assert v <= 1.0, f"T = 250 inpainting trajectory at {v} x tolerance"
    vx = traj_violation(x.cpu().unsqueeze(0), g["x"].unsqueeze(0), rel=LONG_TRAJ_REL, split=None)
    assert vx <= 1.0 and torch.equal(h.cpu().to(torch.int64), g["h"].to(torch.int64))
    print(f"T=250 rs=1: worst recorded latent at {v:.3f} x tolerance, final x at {vx:.3f}")
    gm.noise_fn, gm.trace = None, None


def test_config5_bf16_inpaint_vs_bf16_emulated_sampler():
    """BASELINE configs[4] arithmetic: bf16 MFMA operands TOGETHER with fixed-fragment inpainting (rs = 1;
    equivariant_diffusion.py:423-513) on the `inpaint_T5` fixture.  Yardstick: the oracle sampler driven by the
    bf16-operand emulation of the network (same noise tape), i.e. what the mode is meant to compute.  Stated
    tolerance: 2e-2 of every step's coordinate / feature magnitude against the emulation (operand rounding flips
    with accumulation order and the trajectory passes through |z| ~ 1e3..1e4), 1e-1 against the reference's fp32
    trajectory; the fixed fragment's atom types are exact."""
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    from ml_conformer_generator_amd.equivariant_diffusion import EquivariantDiffusion, PredefinedNoiseSchedule
    from oracle import diffusion_oracle as DO
    g = load_golden("inpaint_T5.npz")
    sd = sd_for(g)
    nm = g["node_mask"]
    T = int(g["T"])

    class Bf16Sampler(DO.SamplerOracle):
        def phi(self, z, t, node_mask, edge_mask, context):
            return _egnn_dynamics_bf16_emulated(self.sd, t, z, node_mask, edge_mask, context)

    orc = Bf16Sampler(sd, T, noise_fn=TapeNoise(g["noise"]))
    orc.trace = []
    x_emu, h_emu = orc.inpaint(nm, edge_mask_of(nm), g["context"], g["z_known"], g["fixed_mask"], 1, 3)
    emu = torch.stack(orc.trace)
    d = EGNNDynamics(device=DEV)
    d.load_reference_state_dict(sd)
    d.set_precision("bf16")
    gm = EquivariantDiffusion(dynamics=d, in_node_nf=8, timesteps=1000, noise_precision=1e-5)
    gm.gamma = PredefinedNoiseSchedule(timesteps=T, precision=1e-5)
    gm.T = T
    gm.noise_fn = TapeNoise(g["noise"], DEV)
    gm.trace = []
    x, h = gm.inpaint(nm.to(DEV), edge_mask_of(nm).to(DEV), g["context"].to(DEV), g["z_known"], g["fixed_mask"], 1, 3)
    assert gm.noise_fn.pos == g["noise"].numel()                     # same draws, same order as the reference
    zt = torch.stack(gm.trace).cpu()
    v_emu = traj_violation(zt, emu, rel=2e-2)
    v_ref = traj_violation(zt, g["z_trace"], rel=1e-1)
    floor = traj_violation(emu, g["z_trace"], rel=1e-1)               # what bf16 operand rounding itself costs
    assert v_emu <= 1.0, (v_emu, v_ref, floor)
    assert v_ref <= 1.0, (v_emu, v_ref, floor)
    n_f = int(g["fixed_mask"][0].sum())
    assert torch.equal(h.cpu()[:, :n_f].to(torch.int64), g["h"][:, :n_f].to(torch.int64))
    assert bool(torch.isfinite(x).all())


# Stated tolerances of the bf16 operand mode over a WHOLE trajectory at the judged step counts (round 6).  Per kept latent and
# channel group, relative to that latent's own group magnitude in the reference, like every trajectory test here:
#   vs the reference's fp32 trajectory   BF16_TRAJ_REL_REF   (operand rounding of 101 / 501 network calls, compounded by the
#                                                              sampler's own 1/alpha_ts amplification and the contractive weights)
#   vs the bf16-operand emulation         BF16_TRAJ_REL_EMU   (same roundings, different accumulation order / rounding flips)
# The measured deviations are written to gpurun_out/round6_bf16_judged_lengths.txt (committed as profiles/round6_bf16_judged_lengths.txt).
BF16_TRAJ_REL_REF = 5e-3          # measured worst 1.03e-3 (T = 250, step 99), 2.2e-4 (T = 100): profiles/round6_bf16_judged_lengths.txt
BF16_TRAJ_REL_EMU = 1e-3          # measured worst 1.9e-4 / 4.7e-6


def _per_latent_rel_dev(a, b):
    """[(x-group, h-group)] per latent: max|a - b| of the channel group / max|b| of that group in that latent."""
    out = []
    for k in range(b.shape[0]):
        row = []
        for sl in (slice(0, 3), slice(3, None)):
            sc = float(b[k][..., sl].abs().max())
            row.append(float((a[k][..., sl] - b[k][..., sl]).abs().max()) / sc if sc > 0 else 0.0)
        out.append(tuple(row))
    return out


def test_bf16_mode_at_the_judged_step_counts_vs_reference_and_emulation():
    """Round-5 review, weak #1: configs[4]'s arithmetic - bf16 MFMA operands - had met the reference over FIVE steps only
    (`inpaint_T5.npz`); its judged length is T = 250 with resample_steps = 1 = 501 denoiser calls
    (equivariant_diffusion.py:423-513).  Here the two reference-generated fixtures at the judged lengths -
    `e2e_T100_b2n27.npz` (forward, 101 calls) and `inpaint_T250_rs1_b2.npz` (inpainting, 501 calls, 751 noise tensors) - are
    replayed through the HIP sampler in "bf16" mode under the recorded tapes:
      * same number and order of noise draws, same number of sampler steps;
      * EVERY recorded latent (each 10th / 50th step) and the final x within BF16_TRAJ_REL_REF of the REFERENCE trajectory;
      * within BF16_TRAJ_REL_EMU of the oracle sampler driven by the bf16-operand emulation of the network (what the mode is
        meant to compute) - by default at the first three kept latents of each fixture (31 / 152 emulated calls),
        MCG_ORACLE_FULL=1 replays both in full and compares the final x too;
      * atom types: the fixed fragment's exact (inpainting), all of them exact where the reference's own top-2 margin of the
        decoded class exceeds the measured deviation of the features.
    The per-latent deviations (the growth law over 100 / 250 levels) are printed and written to gpurun_out/."""
    import os
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    from ml_conformer_generator_amd.equivariant_diffusion import EquivariantDiffusion, PredefinedNoiseSchedule
    from oracle import diffusion_oracle as DO
    full = os.environ.get("MCG_ORACLE_FULL", "0") == "1"
    lines = ["# bf16 operand mode at the judged step counts: deviation per kept latent, max|a - b| of the channel group / max|b| of that group",
             "# (x = coordinates, channels 0..2; h = atom-type features, channels 3..); tests/test_hip_parity.py::"
             "test_bf16_mode_at_the_judged_step_counts_vs_reference_and_emulation",
             f"# stated tolerances: {BF16_TRAJ_REL_REF:g} vs the reference's fp32 trajectory, {BF16_TRAJ_REL_EMU:g} vs the bf16-operand emulation"]
    failures = []

    class _Enough(Exception):
        pass

    for name, kind in (("e2e_T100_b2n27.npz", "forward"), ("inpaint_T250_rs1_b2.npz", "inpaint")):
        g = load_golden(name)
        sd = sd_for(g)
        nm = g["node_mask"]
        T = int(g["T"])
        idx = g["z_trace_index"].long()
        n_steps = int(g["n_sampler_steps"]) if "n_sampler_steps" in g else T

        def run_hip(mode):
            d = EGNNDynamics(device=DEV)
            d.load_reference_state_dict(sd)
            d.set_precision(mode)
            gm = EquivariantDiffusion(dynamics=d, in_node_nf=8, timesteps=1000, noise_precision=1e-5)
            gm.gamma = PredefinedNoiseSchedule(timesteps=T, precision=1e-5)
            gm.T = T
            gm.noise_fn = TapeNoise(g["noise"], DEV)
            gm.trace = []
            if kind == "forward":
                x, h = gm(nm.to(DEV), edge_mask_of(nm).to(DEV), g["context"].to(DEV), 0)
            else:
                x, h = gm.inpaint(nm.to(DEV), edge_mask_of(nm).to(DEV), g["context"].to(DEV), g["z_known"], g["fixed_mask"], 1, 3)
            assert gm.noise_fn.pos == g["noise"].numel() and len(gm.trace) == n_steps     # same draws, same order, same length
            return torch.stack(gm.trace).cpu()[idx], x.cpu(), h.cpu()

        z16, x16, h16 = run_hip("bf16")
        z32, x32, h32 = run_hip("f32")
        # the emulation: oracle sampler with the network's hidden-size contractions on bf16-rounded operands
        keep = idx.numel() if full else 3
        stop_after = None if full else int(idx[keep - 1]) + 1

        class _Trace(list):
            def append(self, z):
                super().append(z)
                if stop_after is not None and len(self) >= stop_after:
                    raise _Enough

        class Bf16Sampler(DO.SamplerOracle):
            def phi(self, z, t, node_mask, edge_mask, context):
                return _egnn_dynamics_bf16_emulated(self.sd, t, z, node_mask, edge_mask, context)

        orc = Bf16Sampler(sd, T, noise_fn=TapeNoise(g["noise"]))
        orc.trace = _Trace()
        x_emu = None
        try:
            if kind == "forward":
                x_emu, _ = orc.forward(nm, edge_mask_of(nm), g["context"], 0)
            else:
                x_emu, _ = orc.inpaint(nm, edge_mask_of(nm), g["context"], g["z_known"], g["fixed_mask"], 1, 3)
        except _Enough:
            pass
        emu = torch.stack(list(orc.trace))[idx[:keep]]

        d_ref = _per_latent_rel_dev(z16, g["z_trace"])
        d_f32 = _per_latent_rel_dev(z32, g["z_trace"])
        d_emu = _per_latent_rel_dev(z16[:keep], emu)
        d_floor = _per_latent_rel_dev(emu, g["z_trace"][:keep])        # what bf16 operand rounding itself costs (emulation vs fp32 reference)
        lines.append(f"\n## {name}: {kind}, T = {T}, {n_steps} sampler steps, B = {nm.shape[0]}, kept latents at steps {idx.tolist()}")
        lines.append(f"# {'step':>5} {'bf16 vs ref x':>14} {'bf16 vs ref h':>14} {'f32 vs ref x':>13} {'f32 vs ref h':>13} "
                     f"{'bf16 vs emu x':>14} {'bf16 vs emu h':>14} {'emu vs ref x':>13} {'emu vs ref h':>13}")
        for k in range(idx.numel()):
            e = ("%14.3e %14.3e %13.3e %13.3e" % (d_emu[k] + d_floor[k])) if k < keep else ("%14s %14s %13s %13s" % ("-", "-", "-", "-"))
            lines.append("  %5d %14.3e %14.3e %13.3e %13.3e %s" % (int(idx[k]), d_ref[k][0], d_ref[k][1], d_f32[k][0], d_f32[k][1], e))
        sx = float(g["x"].abs().max())
        dx_ref = float((x16 - g["x"]).abs().max()) / sx
        dx_f32 = float((x32 - g["x"]).abs().max()) / sx
        line = f"  final x: bf16 vs ref {dx_ref:.3e}, f32 vs ref {dx_f32:.3e}"
        if x_emu is not None:
            dx_emu = float((x16 - x_emu).abs().max()) / sx
            line += f", bf16 vs emulation {dx_emu:.3e}, emulation vs ref {float((x_emu - g['x']).abs().max()) / sx:.3e}"
            if dx_emu > BF16_TRAJ_REL_EMU:
                failures.append(f"{name}: final x at {dx_emu:.3e} of max|x| from the emulation")
        lines.append(line)
        # atom types: decoded from z0's features (argmax over 7 of 8 classes, equivariant_diffusion.py:261-285)
        h_ref = g["h"].to(torch.int64)
        same = (h16.to(torch.int64) == h_ref).all(dim=2)
        real = nm[:, :, 0] > 0
        n_diff = int((~same & real).sum())
        lines.append(f"  atom types: {n_diff} of {int(real.sum())} real atoms decode differently from the reference in bf16 mode "
                     f"(f32 mode: {int((~(h32.to(torch.int64) == h_ref).all(dim=2) & real).sum())})")
        if kind == "inpaint":
            n_f = int(g["fixed_mask"][0].sum())
            if not torch.equal(h16[:, :n_f].to(torch.int64), h_ref[:, :n_f]):
                failures.append(f"{name}: the fixed fragment's atom types differ")
        if n_diff:
            failures.append(f"{name}: {n_diff} atom types differ from the reference")
        worst_ref = max(max(r) for r in d_ref)
        worst_emu = max(max(r) for r in d_emu)
        lines.append(f"  worst kept latent: {worst_ref:.3e} vs the reference (tolerance {BF16_TRAJ_REL_REF:g}), {worst_emu:.3e} vs the "
                     f"emulation over the first {keep} (tolerance {BF16_TRAJ_REL_EMU:g}); f32 mode {max(max(r) for r in d_f32):.3e}")
        if worst_ref > BF16_TRAJ_REL_REF or dx_ref > BF16_TRAJ_REL_REF:
            failures.append(f"{name}: {max(worst_ref, dx_ref):.3e} from the reference trajectory")
        if worst_emu > BF16_TRAJ_REL_EMU:
            failures.append(f"{name}: {worst_emu:.3e} from the bf16 emulation")
        if not bool(torch.isfinite(x16).all()):
            failures.append(f"{name}: non-finite output")
    report = "\n".join(lines) + "\n"
    print(report)
    try:
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        with open(os.path.join(REPO, "gpurun_out", "round6_bf16_judged_lengths" + ("_full" if full else "") + ".txt"), "w") as f:
            f.write(report)
    except OSError:
        pass
    assert not failures, failures


def test_config5_share_ragged256_bf16_inpaint_properties():
    """One GPU's share of BASELINE configs[4] at full width AND full length (round 6: T = 250 -> 501 denoiser calls, ~2 s per
    run; T = 20 until round 5): 256 ragged molecules (15..39 atoms), bf16 operands, fixed 8-atom fragment, resample_steps = 1
    (the oracle would need ~10 min per call at this size).  Size-independent properties: finite outputs, one-hot atom types
    with the reference's 7-of-8 argmax, padded slots exactly zero, bit-identical reruns (no atomics anywhere), 64-row units
    and the fused node launch in use."""
    from ml_conformer_generator_amd import MLConformerGenerator
    from ml_conformer_generator_amd import weights as W
    from ml_conformer_generator_amd.synthetic import DUMMY_CONTEXT
    gen = MLConformerGenerator(diffusion_steps=250, device=DEV, edm_weights=W.synth_edm_state_dict(1234, weight_gain=0.3),
                               adj_mat_seer_weights=W.synth_adj_mat_seer_state_dict(4321), compute_dtype="bf16")
    fx = torch.tensor([[1.25 * i, 0.72 * (i % 2), 0.3 * ((i // 2) % 2)] for i in range(8)], dtype=torch.float32)
    frag = (fx - fx.mean(0), [6, 6, 6, 6, 6, 6, 17, 17])
    ctx = torch.tensor(DUMMY_CONTEXT)

    def run():
        torch.manual_seed(3)
        return gen.edm_tensors(ctx, n_samples=256, min_n_nodes=15, max_n_nodes=39, resample_steps=1, fixed_fragment=frag,
                               inertial_fragment_matching=False, blend_power=3)
    from ml_conformer_generator_amd import _lib
    c = np.zeros(32, dtype=np.int64)
    _lib.check(_lib.lib().mcg_debug_gemm_launches(c.ctypes.data, 1), "mcg_debug_gemm_launches")        # reset
    x1, h1, nm1 = run()
    _lib.check(_lib.lib().mcg_debug_gemm_launches(c.ctypes.data, 1), "mcg_debug_gemm_launches")        # what the first call's capture issued
    x2, h2, nm2 = run()
    assert torch.equal(x1, x2) and torch.equal(h1, h2)
    assert bool(torch.isfinite(x1).all())
    assert torch.equal(h1.sum(2, keepdim=True), nm1)                 # exactly one atom class per real atom, none on padding
    assert float(h1[:, :, 7].abs().max()) == 0.0                     # the reference's 7-of-8 argmax: Br is never emitted
    assert float((x1 * (1 - nm1)).abs().max()) == 0.0 and float((h1 * (1 - nm1)).abs().max()) == 0.0
    assert gen.generative_model.dynamics.plan(nm1.sum(1).reshape(-1).to(torch.int32).cpu(), 39).edge_mt == 4
    assert int(c.reshape(4, 8)[1][6]) >= 36        # the two molecule ranges' captures each issued 18 fused node launches


@pytest.mark.parametrize("mode", ["f32", "f32x6"])
def test_merge_fragments_vs_golden(sampler_factory, mode):
    g = load_golden("merge_T10_L10.npz")
    nm = g["node_mask"]
    gm = sampler_factory(int(g["T"]), g, mode)
    gm.noise_fn = TapeNoise(g["noise"], DEV)
    gm.trace = []
    x, h = gm.merge_fragments(nm.to(DEV), edge_mask_of(nm).to(DEV), g["fixed_mask"], g["context"].to(DEV),
                              g["z_known"], int(g["diffusion_level"]), 1, 3)
    assert gm.noise_fn.pos == g["noise"].numel()
    v = traj_violation(torch.stack(gm.trace).cpu(), g["z_trace"])
    assert v <= 1.0, f"trajectory at {v} x tolerance"
    assert traj_violation(x.cpu().unsqueeze(0), g["x"].unsqueeze(0), split=None) <= 1.0
    assert torch.equal(h.cpu().to(torch.int64), g["h"].to(torch.int64))
    gm.noise_fn, gm.trace = None, None
    with pytest.raises(IndexError):      # diffusion_level > T fails like the reference (quirk H5)
        gm.merge_fragments(nm.to(DEV), None, g["fixed_mask"], g["context"].to(DEV), g["z_known"], 50)


@pytest.fixture(scope="module")
def gcn(gcn_sd):
    from ml_conformer_generator_amd.adj_mat_seer import AdjMatSeer
    m = AdjMatSeer(device=DEV)
    m.load_state_dict(gcn_sd)
    return m


def test_adj_mat_seer_vs_golden(gcn):
    g = load_golden("adj_mat_seer_b4.npz")
    bond, logits = gcn.bond_orders(g["elements"], g["dist_mat"], g["adj_mat"], with_logits=True)
    ok, err, sc = close(logits, g["logits"], rtol=1e-4, atol=1e-5)
    assert ok, f"logits err {err} scale {sc}"
    logits = logits.cpu()
    assert torch.equal(logits, logits.transpose(1, 2))                       # exactly symmetric
    assert torch.equal(torch.argmax(logits, -1).to(torch.int8), bond.cpu())  # device argmax == argmax of logits
    # adjacency indices: bit-exact wherever the reference's own top-2 margin exceeds the fp32 error bound
    safe = g["margin"] > 1e-3
    assert float(safe.float().mean()) > 0.9
    assert torch.equal(bond.cpu().to(torch.int64)[safe], g["argmax"][safe])
    assert int((bond.cpu().to(torch.int64) != g["argmax"]).sum()) == 0       # and in fact everywhere on this fixture
    gcn.check_inputs_seen()


def test_adj_mat_seer_full_size_properties(gcn):
    """B = 64 at full width: symmetry, batch independence (sample k alone == sample k in batch)."""
    from ml_conformer_generator_amd.synthetic import synth_gcn_inputs
    el, dm, am = synth_gcn_inputs(64, [15 + (i * 7) % 25 for i in range(64)], seed=3)
    bond, logits = gcn.bond_orders(el, dm, am, with_logits=True)
    logits = logits.cpu()
    assert torch.equal(logits, logits.transpose(1, 2))
    one = gcn(el[5:6], dm[5:6], am[5:6]).cpu()
    assert torch.equal(one[0], logits[5])


def test_adj_mat_seer_b64_vs_oracle(gcn, gcn_sd):
    """AdjMatSeer at the BASELINE batch size (B = 64, 15..39 atoms) against the oracle on every logit (rtol 1e-4 /
    atol 1e-5 of max|logits|), bond argmax bit-exact wherever the oracle's top-2 margin exceeds 1e-3 (and the number of
    entries below that margin is reported).  The oracle needs ~0.2 s for this."""
    from ml_conformer_generator_amd.synthetic import synth_gcn_inputs
    from oracle import gcn_oracle as GO
    sizes = [15 + (i * 7) % 25 for i in range(64)]
    el, dm, am = synth_gcn_inputs(64, sizes, seed=3)
    ref = GO.adj_mat_seer(gcn_sd, el, dm, am)
    bond, logits = gcn.bond_orders(el, dm, am, with_logits=True)
    ok, err, sc = close(logits, ref, rtol=1e-4, atol=1e-5)
    assert ok, f"logits err {err} scale {sc}"
    top2 = torch.topk(ref, 2, dim=-1).values
    margin = top2[..., 0] - top2[..., 1]
    safe = margin > 1e-3
    assert float(safe.float().mean()) > 0.9
    got = bond.cpu().to(torch.int64)
    assert torch.equal(got[safe], ref.argmax(-1)[safe])
    n_diff = int((got != ref.argmax(-1)).sum())
    print(f"B=64 GCN: max|logit err| {err:.2e} (scale {sc:.2e}); {int((~safe).sum())} of {safe.numel()} entries below the 1e-3 "
          f"margin; argmax differs on {n_diff} entries in total")
    assert n_diff <= int((~safe).sum())


@pytest.mark.parametrize("name", ["e2e_T20_b4n19.npz", "e2e_T8_b8n27.npz", "e2e_merge_T10_L10.npz"])
def test_generate_path_end_to_end_vs_reference_golden(sampler_factory, gcn, gcn_sd, name):
    """north_star's parity clause on the COMPOSED path (conformer_generator.py:330-366; mol_utils.py:146-194,210-211):
    HIP sampler under the reference's recorded noise tape -> mcg_handoff -> mcg_gcn_forward -> bond argmax ->
    mcg_bond_writeback, against what the reference's own EquivariantDiffusion and AdjMatSeer produce from that tape
    (tools/make_golden.py section 8; RDKit's two decisions replaced by the labelled substitutes of
    oracle/host_oracle.py:adj_mat_seer_input on BOTH sides).
    Stated: atom types and elements exact; x within the trajectory tolerance (1e-3 of max|x|); distances follow x;
    connectivity input exact except pairs whose distance sits within the x error of its threshold; logits within 1e-3 of
    max|logits|; adjacency argmax (bond orders) EQUAL on every entry whose reference top-2 margin exceeds 100x the
    measured logit error - the excluded share is printed and must stay below 1 %; the mirrored lower-triangle write-back
    equals the oracle's write-back of the reference argmax on those entries."""
    from ml_conformer_generator_amd.handoff import bond_writeback_hip, prepare_adj_mat_seer_input_hip
    from oracle import host_oracle as HO
    g = load_golden(name)
    nm = g["node_mask"]
    gm = sampler_factory(int(g["T"]), g, "f32")
    gm.noise_fn = TapeNoise(g["noise"], DEV)
    merge = "route" in g            # the fragment-merge route of edm_samples (conformer_generator.py:231-240)
    if merge:
        x, h = gm.merge_fragments(nm.to(DEV), edge_mask_of(nm).to(DEV), g["fixed_mask"], g["context"].to(DEV), g["z_known"],
                                  int(g["diffusion_level"]), int(g["resample_steps"]), int(g["blend_power"]))
    else:
        x, h = gm(nm.to(DEV), edge_mask_of(nm).to(DEV), g["context"].to(DEV), 0)
    assert gm.noise_fn.pos == g["noise"].numel()
    gm.noise_fn = None
    assert torch.equal(h.cpu().to(torch.int64), g["h"].to(torch.int64))                       # atom types exact
    vx = traj_violation(x.cpu().unsqueeze(0), g["x"].unsqueeze(0), split=None)
    assert vx <= 1.0, f"x at {vx} x tolerance"
    x_err = float((x.cpu() - g["x"]).abs().max())
    n_nodes = g["n_nodes"]
    el, dm, am = prepare_adj_mat_seer_input_hip(x, h, n_nodes.to(DEV))
    assert torch.equal(el.cpu(), g["elements"])
    d_err = float((dm.cpu() - g["dist_mat"]).abs().max())
    assert d_err <= 4.0 * x_err + 1e-5, (d_err, x_err)
    rc = torch.zeros(36)
    for z, r in HO._RCOV.items():
        rc[z] = r
    thr = 1.3 * (rc[g["elements"]].unsqueeze(1) + rc[g["elements"]].unsqueeze(2))
    borderline = ((g["dist_mat"] - thr).abs() <= 4.0 * x_err + 1e-5) & (thr > 0)
    assert torch.equal(am.cpu()[~borderline], g["adj_mat"][~borderline])
    am_equal = torch.equal(am.cpu(), g["adj_mat"])
    assert am_equal, f"connectivity input differs on {int((am.cpu() != g['adj_mat']).sum())} borderline pairs"
    bond, logits = gcn.bond_orders(el, dm, am, with_logits=True)
    l_err = float((logits.cpu() - g["logits"]).abs().max())
    l_sc = float(g["logits"].abs().max())
    assert l_err <= 1e-3 * l_sc, (l_err, l_sc)
    D = g["elements"].shape[1]
    inside = (torch.arange(D).view(1, D, 1) < n_nodes.view(-1, 1, 1)) & (torch.arange(D).view(1, 1, D) < n_nodes.view(-1, 1, 1))
    safe = g["margin"] > 100.0 * l_err
    got = bond.cpu().to(torch.int64)
    assert torch.equal(got[safe], g["argmax"][safe])                                         # adjacency indices bit-exact
    excluded = float((~safe & inside).sum()) / float(inside.sum())
    n_diff = int(((got != g["argmax"]) & inside).sum())
    print(f"{name}: x err {x_err:.2e} (max|x| {float(g['x'].abs().max()):.1f}), dist err {d_err:.2e}, logit err {l_err:.2e} "
          f"(scale {l_sc:.2e}); {100 * excluded:.3f} % of the in-molecule entries are below 100x that error; "
          f"bond argmax differs from the reference on {n_diff} in-molecule entries")
    assert excluded < 0.01
    sym, _ = bond_writeback_hip(bond, el, n_nodes.to(DEV))
    sym_ref, _ = HO.bond_writeback(g["argmax"], g["elements"], n_nodes)
    low = torch.tril(torch.ones(D, D, dtype=torch.bool), -1).unsqueeze(0) & inside
    both = low & safe & safe.transpose(1, 2)
    assert torch.equal(sym.cpu()[both], sym_ref[both])
    if n_diff == 0:
        assert torch.equal(sym.cpu(), sym_ref)
    # the same chain against the ORACLE pipeline driven by the same tape (sampler -> hand-off -> GCN), every stage
    from oracle import diffusion_oracle as DO
    from oracle import gcn_oracle as GO
    orc = DO.SamplerOracle(sd_for(g), int(g["T"]), noise_fn=TapeNoise(g["noise"]))
    if merge:
        xo, ho = orc.merge_fragments(nm, edge_mask_of(nm), g["fixed_mask"], g["context"], g["z_known"], int(g["diffusion_level"]),
                                     int(g["resample_steps"]), int(g["blend_power"]))
    else:
        xo, ho = orc.forward(nm, edge_mask_of(nm), g["context"], 0)
    elo, dmo, amo = HO.adj_mat_seer_input(xo, ho, n_nodes)
    lo = GO.adj_mat_seer(gcn_sd, elo, dmo, amo)
    assert torch.equal(el.cpu(), elo) and torch.equal(am.cpu(), amo)
    assert float((logits.cpu() - lo).abs().max()) <= 1e-3 * l_sc
    safe_o = (torch.topk(lo, 2, dim=-1).values[..., 0] - torch.topk(lo, 2, dim=-1).values[..., 1]) > 100.0 * l_err
    assert torch.equal(got[safe_o], lo.argmax(-1)[safe_o])


def test_aggregate_standalone(edm_sd):
    from ml_conformer_generator_amd import _lib
    from oracle.egnn_oracle import aggregate_standalone
    torch.manual_seed(2)
    n_nodes = torch.tensor([15, 27, 39, 20])
    rows = int((n_nodes * (n_nodes - 1)).sum())
    D = 420
    m = torch.randn(rows, D)
    gate = torch.rand(rows)
    ref = aggregate_standalone(m, gate, n_nodes)
    first, cnt, off = [], [], 0
    for n in n_nodes.tolist():
        for i in range(n):
            first.append(off + i * (n - 1))
            cnt.append(n - 1)
        off += n * (n - 1)
    first = torch.tensor(first, dtype=torch.int32, device=DEV)
    cnt = torch.tensor(cnt, dtype=torch.int32, device=DEV)
    out = torch.empty(int(n_nodes.sum()), D, device=DEV)
    md, gd = m.to(DEV), gate.to(DEV)
    _lib.check(_lib.lib().mcg_egnn_aggregate(md.data_ptr(), gd.data_ptr(), first.data_ptr(), cnt.data_ptr(),
                                             out.data_ptr(), out.shape[0], D, _lib.current_stream_ptr(DEV)), "agg")
    assert torch.allclose(out.cpu(), ref, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------------------- full-size properties (config 2)
def _c2_inputs(B=64, n=27, seed=0):
    g = torch.Generator().manual_seed(seed)
    nm = torch.ones(B, n, 1)
    z = torch.randn(B, n, 11, generator=g)
    z[:, :, :3] -= z[:, :, :3].mean(1, keepdim=True)
    ctx = torch.tensor([-0.99, -1.66, -1.66]).view(1, 1, 3).repeat(B, n, 1)
    t = torch.full((B,), 0.5)
    return nm, z, ctx, t


def test_full_size_equivariance_determinism_padding(dyn):
    nm, z, ctx, t = _c2_inputs()
    B, n, _ = z.shape
    plan = dyn.plan(torch.full((B,), n, dtype=torch.int32), n)
    out1 = dyn.run(plan, t.to(DEV), z.to(DEV), ctx.to(DEV)).cpu()
    out2 = dyn.run(plan, t.to(DEV), z.to(DEV), ctx.to(DEV)).cpu()
    assert torch.equal(out1, out2)                                   # deterministic (no atomics)
    assert torch.isfinite(out1).all()
    # E(3) equivariance: rotate + translate the coordinates -> velocities rotate, features invariant
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=torch.Generator().manual_seed(1)))
    zr = z.clone()
    zr[:, :, :3] = z[:, :, :3] @ q.T + torch.tensor([0.3, -1.2, 2.0])
    outr = dyn.run(plan, t.to(DEV), zr.to(DEV), ctx.to(DEV)).cpu()
    sc = float(out1.abs().max())
    assert float((outr[:, :, :3] - out1[:, :, :3] @ q.T).abs().max()) < 2e-4 * sc
    assert float((outr[:, :, 3:] - out1[:, :, 3:]).abs().max()) < 2e-4 * sc
    # permutation of atoms inside every molecule
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(2))
    outp = dyn.run(plan, t.to(DEV), z[:, perm].contiguous().to(DEV), ctx.to(DEV)).cpu()
    assert float((outp - out1[:, perm]).abs().max()) < 2e-4 * sc
    # pad-width independence: same molecules in a wider padded layout
    N2 = 35
    zp = torch.zeros(B, N2, 11)
    zp[:, :n] = z
    cp = torch.zeros(B, N2, 3)
    cp[:, :n] = ctx
    plan2 = dyn.plan(torch.full((B,), n, dtype=torch.int32), N2)
    outw = dyn.run(plan2, t.to(DEV), zp.to(DEV), cp.to(DEV)).cpu()
    assert torch.equal(outw[:, :n], out1) and float(outw[:, n:].abs().max()) == 0.0
    # batch independence: molecule 7 alone
    plan1 = dyn.plan(torch.tensor([n], dtype=torch.int32), n)
    o7 = dyn.run(plan1, t[:1].to(DEV), z[7:8].contiguous().to(DEV), ctx[7:8].contiguous().to(DEV)).cpu()
    assert float((o7[0] - out1[7]).abs().max()) < 1e-5 * sc


def test_generator_end_to_end_c1(edm_sd, gcn_sd):
    """Config C1 plumbing on the HIP path: ceyyag heavy atoms, n_samples=4, 20 steps."""
    from ml_conformer_generator_amd import MLConformerGenerator
    g = load_golden("context_shape.npz")
    gen = MLConformerGenerator(diffusion_steps=20, device=DEV, edm_weights=edm_sd, adj_mat_seer_weights=gcn_sd)
    torch.manual_seed(0)
    mols = gen.generate_conformers(reference_conformer=g["ceyyag_xyz"], n_samples=4, variance=2)
    lb = gen.last_batch
    assert lb["x"].shape[0] == 4 and 15 <= int(lb["n_nodes"].min()) and int(lb["n_nodes"].max()) <= 19
    assert torch.isfinite(lb["x"]).all()
    assert len(mols) <= 4
    with pytest.raises(ValueError):
        gen.generate_conformers(reference_context=torch.tensor([50.0, 100.0, 130.0]))
    with pytest.raises(ValueError):
        gen.generate_conformers()


@pytest.mark.parametrize("ifm", [False, True])
def test_generator_fixed_fragment_modes(gcn_sd, ifm):
    """Plumbing of both fixed-fragment strategies (inpaint / inertial fragment matching + merge) on the HIP path:
    the fragment's atom types survive where the reference pins them, shapes and counts are right."""
    from ml_conformer_generator_amd import MLConformerGenerator
    g = load_golden("ifm_front_end.npz")
    # property test (shapes, finiteness, API flow): contractive legacy weights - an UNTRAINED denoiser with the
    # mutation-checked "v2" gains overflows fp32 under resampling (also in the reference; see weights.py)
    from ml_conformer_generator_amd import weights as W
    gen = MLConformerGenerator(diffusion_steps=12, device=DEV, edm_weights=W.synth_edm_state_dict(1234, weight_gain=0.3),
                               adj_mat_seer_weights=gcn_sd)
    frag = (g["frag_x"], g["frag_z"].tolist())
    torch.manual_seed(3)
    x, h, nm = gen.edm_tensors(g["ref_context"], n_samples=3, min_n_nodes=21, max_n_nodes=25, resample_steps=1,
                               fixed_fragment=frag, inertial_fragment_matching=ifm, blend_power=3,
                               ifm_diffusion_level=5)
    assert x.shape == (3, 25, 3) and h.shape == (3, 25, 8) and torch.isfinite(x).all()
    n = nm.sum(1).reshape(-1)
    assert int(n.min()) >= 21 and int(n.max()) <= 25
    assert float((h.sum(2) - nm.squeeze(2)).abs().max()) == 0.0            # one-hot on real atoms, zero on padding
    with pytest.raises(IndexError):                                         # level > steps, as the reference
        gen.edm_tensors(g["ref_context"], n_samples=2, min_n_nodes=21, max_n_nodes=25, fixed_fragment=frag,
                        inertial_fragment_matching=True, ifm_diffusion_level=50)


# ------------------------------------------------------------------------------- bf16 operand mode (configs[4])
def _bf(x):
    return x.to(torch.bfloat16).to(torch.float32)


def _egnn_dynamics_bf16_emulated(sd, t, xh, node_mask, edge_mask, context):
    """fp32 torch emulation of what the bf16 mode computes: the reference network with the OPERANDS of
    every hidden-size contraction rounded to bf16 (fp32 accumulate) and the first edge layer in its
    factorised form - a yardstick that separates 'bf16 rounding' from 'kernel bug'."""
    import torch.nn.functional as F
    from oracle import egnn_oracle as EO
    B, N, _ = xh.shape
    row, col = EO.dense_edge_index(N, B)
    nm = node_mask.reshape(B * N, 1)
    em = edge_mask.reshape(B * N * N, 1)
    flat = xh.reshape(B * N, -1) * nm
    x0 = flat[:, :3].clone()
    h = torch.cat([flat[:, 3:], t.reshape(B, 1).repeat(1, N).reshape(B * N, 1), context.reshape(B * N, -1)], 1)
    p0 = "dynamics.egnn."
    h = F.linear(h, sd[p0 + "embedding.weight"], sd[p0 + "embedding.bias"])
    x = x0.clone()
    d0, _ = EO.pair_geometry(x0, row, col)

    def lin16(a, w, b=None):
        return F.linear(_bf(a), _bf(w), b)

    def edge_mlp(p, key, h, d2):
        w1, b1 = sd[p + key + ".0.weight"], sd[p + key + ".0.bias"]
        pa = lin16(h, w1[:, :420], b1)
        pb = lin16(h, w1[:, 420:840])
        pre = pa[row] + pb[col] + d2 * w1[:, 840] + d0 * w1[:, 841]
        return F.silu(lin16(F.silu(pre), sd[p + key + ".2.weight"], sd[p + key + ".2.bias"]))

    for k in range(9):
        bp = f"{p0}e_block_{k}."
        d2, unit = EO.pair_geometry(x, row, col)
        for gname in ("gcl_0.", "gcl_1."):
            p = bp + gname
            m = edge_mlp(p, "edge_mlp", h, d2)
            gate = torch.sigmoid(F.linear(m, sd[p + "att_mlp.0.weight"], sd[p + "att_mlp.0.bias"]))
            agg = EO.segment_sum(m * gate * em, row, h.size(0))
            hid = F.silu(lin16(torch.cat([h, agg], 1), sd[p + "node_mlp.0.weight"], sd[p + "node_mlp.0.bias"]))
            h = (h + lin16(hid, sd[p + "node_mlp.2.weight"], sd[p + "node_mlp.2.bias"])) * nm
        p = bp + "gcl_equiv."
        m = edge_mlp(p, "coord_mlp", h, d2)
        phi = F.linear(m, sd[p + "coord_mlp.4.weight"])
        x = (x + EO.segment_sum(unit * phi * em, row, x.size(0))) * nm
    h = F.linear(h, sd[p0 + "embedding_out.weight"], sd[p0 + "embedding_out.bias"]) * nm
    vel = EO.masked_mean_removal(((x - x0) * nm).reshape(B, N, 3), node_mask.reshape(B, N, 1))
    return torch.cat([vel, h[:, :8].reshape(B, N, -1)], dim=2)


def test_bf16_mode_vs_emulation_and_fp32(edm_sd):
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    from oracle import egnn_oracle as EO
    from oracle import host_oracle as HO
    d = EGNNDynamics(device=DEV)
    d.load_reference_state_dict(edm_sd)
    d.set_precision("bf16")
    torch.manual_seed(11)
    sizes = torch.tensor([19, 27, 15, 33])
    N = 33
    nm, em = HO.masks_from_sizes(sizes, N)
    z = torch.randn(4, N, 11) * nm
    ctx = torch.randn(4, 1, 3).repeat(1, N, 1) * nm
    t = torch.full((4, 1), 0.4)
    out = d(t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV)).cpu()
    assert next(iter(d._plans.values())).edge_mt == 4                  # 64-row-tile bf16 kernel
    out_mt1 = d.run(d.plan(sizes, N, edge_mt=1), t.reshape(-1).to(DEV), z.to(DEV), ctx.to(DEV)).cpu()   # 16-row-tile kernel
    emu = _egnn_dynamics_bf16_emulated(edm_sd, t, z, nm, em, ctx)
    assert float((out_mt1 - emu).abs().max()) / float(emu.abs().max()) < 3e-3
    ref = EO.egnn_dynamics(edm_sd, t, z, nm, em, ctx)
    sc = float(ref.abs().max())
    e_emu = float((out - emu).abs().max()) / sc
    e_ref = float((out - ref).abs().max()) / sc
    e_floor = float((emu - ref).abs().max()) / sc            # what bf16 operand rounding itself costs
    # stated bf16 tolerance: 3e-3 of max|out| vs the bf16-operand emulation (accumulation-order +
    # rounding-boundary effects only), 3e-2 vs the fp32 reference
    assert e_emu < 3e-3, (e_emu, e_ref, e_floor)
    assert e_ref < 3e-2, (e_emu, e_ref, e_floor)
    assert float((out * (1 - nm)).abs().max()) == 0.0
    d.set_precision("f32")
    out32 = d(t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV)).cpu()
    assert float((out32 - ref).abs().max()) / sc < 1e-5


def test_handoff_kernel_vs_oracle():
    """f1: mcg_handoff against the ORACLE restatement of the tensor half of samples_to_rdkit_mol +
    prepare_adj_mat_seer_input (mol_utils.py:18-57,146-194; fp64 distances of the "%.9f" coordinates, + I, pad 42;
    connectivity substitute + I).  Round 5: the kernel computes what the reference computes - the "%.9f" text round trip of
    the coordinate (exact in fp64), fp64 `distance_matrix`, ONE rounding to fp32 - so elements, distances AND the
    covalent-rule adjacency are BIT-EXACT, no borderline carve-out.  Also against the reference's own `distance_matrix`
    (fixture `handoff_tensor_half.npz`)."""
    from ml_conformer_generator_amd.handoff import prepare_adj_mat_seer_input_hip
    from oracle import host_oracle as HO
    torch.manual_seed(4)
    B, N = 9, 27
    n_nodes = torch.randint(15, 28, (B,))
    n_nodes[0] = 27
    real = (torch.arange(N).unsqueeze(0) < n_nodes.unsqueeze(1)).float().unsqueeze(2)
    x = torch.cumsum(torch.nn.functional.normalize(torch.randn(B, N, 3), dim=2) * 1.45, dim=1) * real
    h = torch.nn.functional.one_hot(torch.randint(0, 7, (B, N)), 8).float() * real
    el0, dm0, am0 = HO.adj_mat_seer_input(x, h, n_nodes)
    el1, dm1, am1 = prepare_adj_mat_seer_input_hip(x.to(DEV), h.to(DEV), n_nodes)
    assert torch.equal(el1.cpu(), el0)
    n_diff = int((dm1.cpu() != dm0).sum())
    assert n_diff == 0, f"{n_diff} of {dm0.numel()} distances differ from the fp64 -> fp32 reference arithmetic"
    assert torch.equal(am1.cpu(), am0)
    assert int(am0.sum()) > B * 42                                   # some bonds were actually perceived
    # coordinates at other magnitudes (the "%.9f" rounding acts on the 9th DECIMAL: its effect grows as |x| shrinks)
    for scale in (1e-3, 0.37, 11.0, 250.0):
        el0, dm0, am0 = HO.adj_mat_seer_input(x * scale, h, n_nodes)
        el1, dm1, am1 = prepare_adj_mat_seer_input_hip((x * scale).to(DEV), h.to(DEV), n_nodes)
        assert torch.equal(dm1.cpu(), dm0) and torch.equal(am1.cpu(), am0), scale
    # the reference's own arithmetic: `distance_matrix` on the text-round-tripped coordinates (tools/make_golden.py)
    g = load_golden("handoff_tensor_half.npz")
    el1, dm1, am1 = prepare_adj_mat_seer_input_hip(g["x"].to(DEV), g["h"].to(DEV), g["n_nodes"])
    assert torch.equal(el1.cpu(), g["elements"]) and torch.equal(dm1.cpu(), g["dist_mat"]) and torch.equal(am1.cpu(), g["adj_mat"])
    el1, dm1, am1 = prepare_adj_mat_seer_input_hip(g["x"].to(DEV), g["h"].to(DEV), g["n_nodes"], connectivity=list(g["conn_cov"]))
    assert torch.equal(dm1.cpu(), g["dist_mat"]) and torch.equal(am1.cpu(), g["adj_mat"])


def test_handoff_ex_random_orders_and_connectivities_vs_oracle(gcn, gcn_sd):
    """f1: `mcg_handoff_ex` with an externally supplied atom order and connectivity (the two RDKit-owned decisions of
    `canonicalise`, mol_utils.py:110-126) against `host_oracle.adj_mat_seer_input(order=, conn=)`: random permutations
    and random symmetric connectivities on ragged molecules (incl. n = 1, 42).  Elements, adjacency and (round 5: fp64
    arithmetic in the kernel) distances BIT-EXACT, permuted coordinates exact, GCN logits within 1e-4 of max|logits| of the
    oracle GCN, bond argmax equal wherever the oracle's top-2 margin exceeds 100x the logit error."""
    from ml_conformer_generator_amd import _lib
    from ml_conformer_generator_amd.handoff import prepare_adj_mat_seer_input_hip
    from oracle import gcn_oracle as GO
    from oracle import host_oracle as HO
    g = torch.Generator().manual_seed(41)
    B, N = 12, 42
    n_nodes = torch.randint(15, 40, (B,), generator=g)
    n_nodes[0], n_nodes[1], n_nodes[2] = 42, 1, 2
    real = (torch.arange(N).unsqueeze(0) < n_nodes.unsqueeze(1)).float().unsqueeze(2)
    x = torch.cumsum(torch.nn.functional.normalize(torch.randn(B, N, 3, generator=g), dim=2) * 1.45, dim=1) * real
    h = torch.nn.functional.one_hot(torch.randint(0, 8, (B, N), generator=g), 8).float() * real
    order, conn = [], []
    for b in range(B):
        n = int(n_nodes[b])
        order.append(torch.randperm(n, generator=g).tolist())
        c = torch.rand(n, n, generator=g) < 0.12
        c = (c | c.t()) & ~torch.eye(n, dtype=torch.bool)
        conn.append(c.to(torch.uint8).numpy())
    order[3] = None                                                 # a None row keeps generation order
    for use_order, use_conn in ((True, True), (True, False), (False, True)):
        o, c = (order if use_order else None), (conn if use_conn else None)
        el0, dm0, am0, xc0 = HO.adj_mat_seer_input(x, h, n_nodes, order=o, conn=c, return_coords=True)
        el1, dm1, am1, xo = prepare_adj_mat_seer_input_hip(x.to(DEV), h.to(DEV), n_nodes, order=o, connectivity=c,
                                                           with_coords=True)
        assert torch.equal(el1.cpu(), el0)
        assert torch.equal(dm1.cpu(), dm0) and torch.equal(am1.cpu(), am0)
        for b in range(B):
            n = int(n_nodes[b])
            perm = list(range(n)) if (o is None or o[b] is None) else o[b]
            assert torch.equal(xo.cpu()[b, :n], x[b, perm]) and float(xo.cpu()[b, n:].abs().sum()) == 0.0
        if use_order and use_conn:
            bond, logits = gcn.bond_orders(el1, dm1, am1, with_logits=True)
            lo = GO.adj_mat_seer(gcn_sd, el0, dm0, am0)
            err, sc = float((logits.cpu() - lo).abs().max()), float(lo.abs().max())
            assert err <= 1e-4 * sc, (err, sc)
            top2 = torch.topk(lo, 2, dim=-1).values
            safe = (top2[..., 0] - top2[..., 1]) > 100.0 * err
            assert torch.equal(bond.cpu().to(torch.int64)[safe], lo.argmax(-1)[safe]) and float(safe.float().mean()) > 0.99
    # malformed orders are refused on the host, before any launch
    with pytest.raises(ValueError, match="permutation"):
        prepare_adj_mat_seer_input_hip(x.to(DEV), h.to(DEV), n_nodes, order=[[0] * 42] * B)
    with pytest.raises(ValueError, match="symmetric"):
        bad = [c.copy() for c in conn]
        bad[4][0, 1] ^= 1
        prepare_adj_mat_seer_input_hip(x.to(DEV), h.to(DEV), n_nodes, connectivity=bad)
    # ... and the C ABI reports an out-of-range entry through its flag instead of reading out of bounds
    od = torch.arange(42, dtype=torch.int32).repeat(B, 1)
    od[5, 0] = 77
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    el = torch.empty(B, 42, dtype=torch.long, device=DEV)
    dm = torch.empty(B, 42, 42, device=DEV)
    am = torch.empty(B, 42, 42, device=DEV)
    xd, hd, nd, odd = x.to(DEV), h.to(DEV), n_nodes.to(DEV, torch.int32), od.to(DEV)
    L = _lib.lib()
    _lib.check(L.mcg_handoff_ex(xd.data_ptr(), hd.data_ptr(), nd.data_ptr(), B, N, 1.3, odd.data_ptr(), None, el.data_ptr(),
                                dm.data_ptr(), am.data_ptr(), None, flag.data_ptr(), _lib.current_stream_ptr(DEV)), "mcg_handoff_ex")
    assert int(flag.item()) == 1 and bool(torch.isfinite(dm).all())


def test_generate_path_with_injected_atom_order_vs_reference_golden(sampler_factory, gcn):
    """f1 on the COMPOSED path: HIP sampler under the reference's tape -> `mcg_handoff_ex` with the fixture's fixed
    non-identity atom order and injected connectivity -> `mcg_gcn_forward` -> argmax, against the REFERENCE's AdjMatSeer
    fed the same permuted input built with the reference's own `distance_matrix` (tools/make_golden.py section 9).
    AdjMatSeer is atom-order dependent (the fixture records how many bond entries change with the order), so this pins
    that the product hands the GCN the atoms in the order it is given, and returns coordinates in that order."""
    from ml_conformer_generator_amd.handoff import prepare_adj_mat_seer_input_hip
    g = load_golden("e2e_perm_T20_b4n19.npz")
    assert int(g["order_dependent_entries"]) > 100
    nm = g["node_mask"]
    gm = sampler_factory(int(g["T"]), g, "f32")
    gm.noise_fn = TapeNoise(g["noise"], DEV)
    x, h = gm(nm.to(DEV), edge_mask_of(nm).to(DEV), g["context"].to(DEV), 0)
    gm.noise_fn = None
    assert torch.equal(h.cpu().to(torch.int64), g["h"].to(torch.int64))
    x_err = float((x.cpu() - g["x"]).abs().max())
    n_nodes = g["n_nodes"]
    order, conn = [r.tolist() for r in g["order"]], list(np.asarray(g["conn_in"]))
    el, dm, am, xo = prepare_adj_mat_seer_input_hip(x, h, n_nodes.to(DEV), order=order, connectivity=conn, with_coords=True)
    assert torch.equal(el.cpu(), g["elements"]) and torch.equal(am.cpu(), g["adj_mat"])
    assert float((dm.cpu() - g["dist_mat"]).abs().max()) <= 4.0 * x_err + 1e-5
    assert float((xo.cpu() - g["x_perm"]).abs().max()) <= x_err + 1e-7
    bond, logits = gcn.bond_orders(el, dm, am, with_logits=True)
    l_err, l_sc = float((logits.cpu() - g["logits"]).abs().max()), float(g["logits"].abs().max())
    assert l_err <= 1e-3 * l_sc, (l_err, l_sc)
    D = 42
    inside = (torch.arange(D).view(1, D, 1) < n_nodes.view(-1, 1, 1)) & (torch.arange(D).view(1, 1, D) < n_nodes.view(-1, 1, 1))
    safe = g["margin"] > 100.0 * l_err
    got = bond.cpu().to(torch.int64)
    assert torch.equal(got[safe], g["argmax"][safe])
    assert float((~safe & inside).sum()) / float(inside.sum()) < 0.01
    print(f"e2e_perm: logit err {l_err:.2e} (scale {l_sc:.2e}); bond argmax differs from the reference on "
          f"{int(((got != g['argmax']) & inside).sum())} in-molecule entries")
    # the same sampler output in GENERATION order gives different bonds: the order matters and is honoured
    el_g, dm_g, am_g = prepare_adj_mat_seer_input_hip(x, h, n_nodes.to(DEV), connectivity=conn)
    bond_g = gcn.bond_orders(el_g, dm_g, am_g).cpu().to(torch.int64)
    moved = 0
    for b in range(el.shape[0]):
        n = int(n_nodes[b])
        inv = torch.argsort(torch.tensor(order[b][:n]))
        moved += int((got[b][:n, :n][inv][:, inv] != bond_g[b][:n, :n]).sum())
    assert moved > 100


def test_generator_uses_an_injected_atom_order_provider(edm_sd, gcn_sd):
    """`MLConformerGenerator(atom_order_provider=...)`: the provider sees every molecule in generation order (atomic
    numbers + fp64 coordinates), its order / connectivity reach the GCN, the returned records carry atoms, coordinates and
    bonds in the PROVIDER's order, and a molecule the provider cannot build is dropped (the reference drops a
    `MolFromXYZBlock` failure, mol_utils.py:53-55)."""
    import numpy as np
    from ml_conformer_generator_amd import MLConformerGenerator
    from ml_conformer_generator_amd.handoff import molecules_from_tensors
    seen = []

    def reversing_provider(z, coords):
        seen.append((list(z), np.array(coords)))
        if len(seen) == 2:
            return None                                            # "could not be built"
        n = len(z)
        d = np.linalg.norm(coords[:, None, :] - coords[None, :, :], axis=2)
        return list(range(n - 1, -1, -1)), ((d < 1.9) & ~np.eye(n, dtype=bool)).astype(np.uint8)

    gen = MLConformerGenerator(diffusion_steps=4, device=DEV, edm_weights=edm_sd,
                               adj_mat_seer_weights=gcn_sd, atom_order_provider=reversing_provider)
    ctx = torch.tensor([53.6424, 108.3042, 151.4399])
    torch.manual_seed(3)
    res = gen._generate_shard(ctx, 17, 2, None, 5, 0, None, True, 3, 50)
    lb = gen.last_batch
    assert len(seen) == 5
    n_nodes = lb["n_nodes"].cpu()
    mols = molecules_from_tensors(res["x"], res["elements"], res["bond"], res["n_nodes"], res["valid"])
    assert not mols[1].valid
    zt = torch.tensor([6, 7, 8, 9, 15, 16, 17, 35])
    for b in (0, 2, 3, 4):
        n = int(n_nodes[b])
        z_gen = zt[lb["h"][b, :n].argmax(1).cpu()].tolist()
        assert seen[b][0] == z_gen and np.allclose(seen[b][1], lb["x"][b, :n].cpu().double().numpy())
        assert mols[b].atomic_numbers == z_gen[::-1]
        assert torch.equal(mols[b].coords, lb["x"][b, :n].cpu().flip(0))
    # generation order + covalent rule when there is no provider (no RDKit here): the documented substitutes
    gen2 = MLConformerGenerator(diffusion_steps=4, device=DEV, edm_weights=edm_sd, adj_mat_seer_weights=gcn_sd)
    assert gen2.atom_order_provider is None
    torch.manual_seed(3)
    res2 = gen2._generate_shard(ctx, 17, 2, None, 5, 0, None, True, 3, 50)
    assert torch.equal(res2["x"], gen2.last_batch["x"])


def test_generator_pipelines_pooled_host_stages_like_the_serial_path(edm_sd, gcn_sd):
    """f2 (round 5): the two host stages around the GCN - order + connectivity before it, the finish behind it - fanned
    out over `host_pool` worker processes and pipelined per group of molecules (`_generate_shard`), with fake chunk
    functions standing in for RDKit (tests/fake_host_tasks.py).  Pooled (4 workers), serial (0) and a ONE-launch
    hand-off + GCN over the whole batch with the same decisions must agree bit for bit; molecule 1 cannot be "built" and
    is dropped; the finish sees the records in the provider's order."""
    import os
    import numpy as np
    from ml_conformer_generator_amd import MLConformerGenerator
    from ml_conformer_generator_amd import host_pool as HP
    from ml_conformer_generator_amd import rdkit_order as RO
    from ml_conformer_generator_amd.handoff import bond_writeback_hip, prepare_adj_mat_seer_input_hip
    fake = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_host_tasks.py")
    ctx = torch.tensor([53.6424, 108.3042, 151.4399])
    runs = {}
    for workers in (0, 4):
        gen = MLConformerGenerator(diffusion_steps=4, device=DEV, edm_weights=edm_sd, adj_mat_seer_weights=gcn_sd,
                                   atom_order_provider=HP.TaskRef(fake, "order_chunk_no_sleep"),
                                   finisher=HP.TaskRef(fake, "finish_chunk"), n_host_workers=workers)
        torch.manual_seed(11); torch.cuda.manual_seed(11)
        kept = gen.generate_conformers(reference_context=ctx, n_atoms=17, variance=2, n_samples=40, optimise_geometry=False)
        assert gen.last_host_order_ms is not None and gen.last_host_finish_ms is not None
        runs[workers] = (kept, {k: v.clone() for k, v in gen.last_batch.items()}, gen.last_order, gen.last_valid_fraction)
    (k0, b0, o0, f0), (k4, b4, o4, f4) = runs[0], runs[4]
    assert k0 == k4 and o0 == o4 and f0 == f4 and len(k0) > 0
    for key in b0:
        assert torch.equal(b0[key], b4[key]), key
    assert all(r["mmff"] is False for r in k4)
    # the four group launches against ONE launch over the whole batch with the same decisions
    x, h, n_nodes = b4["x"], b4["h"], b4["n_nodes"]
    order, conn, built = RO.batch_order_and_connectivity(HP.TaskRef(fake, "order_chunk_no_sleep"), x, h, n_nodes)
    assert order == o4 and all(built)
    el, dm, am, xo = prepare_adj_mat_seer_input_hip(x, h, n_nodes, order=order, connectivity=conn, with_coords=True)
    bond = gen.adj_mat_seer.bond_orders(el, dm, am)
    assert torch.equal(el, b4["elements"]) and torch.equal(bond, b4["bond"]) and torch.equal(xo, b4["x_ordered"])
    # what the finish saw: atoms in the provider's order, fp32 coordinates as "%.9f" text would print them
    sym, _ = bond_writeback_hip(bond, el, n_nodes)
    zt = torch.tensor([6, 7, 8, 9, 15, 16, 17, 35])
    done = 0
    for b in range(40):
        n = int(n_nodes[b])
        nb = int((sym[b, :n, :n].cpu().tril(-1) != 0).sum())
        if nb == 0:
            continue
        r = k4[done]; done += 1
        z_gen = zt[h[b, :n].argmax(1).cpu()].tolist()
        assert list(r["z"]) == [z_gen[i] for i in order[b]] and r["bonds"] == nb
        assert r["xyz"] == "%.9f" % float(np.asarray(xo[b, :n].cpu().tolist(), dtype=np.float64).sum())
    assert done == len(k4)


def test_bond_writeback_kernel_vs_oracle():
    """f2: mcg_bond_writeback (lower-triangle bond write-back of mol_utils.py:210-211 + the validity substitute)
    against the oracle's plain-loop restatement, bit-exact, on chains with random extra bonds, broken chains
    (disconnected), over-valent atoms, aromatic bonds, ring closures, junk in the upper triangle / padding and
    the n = 0 / 1 / 42 edges."""
    from ml_conformer_generator_amd.handoff import assemble_molecules, bond_writeback_hip
    from oracle import host_oracle as HO
    g = torch.Generator().manual_seed(5)
    B, D = 96, 42
    n = torch.randint(3, 40, (B,), generator=g)
    n[0], n[1], n[2] = 0, 1, 42
    el = torch.zeros(B, D, dtype=torch.long)
    bond = torch.randint(0, 5, (B, D, D), generator=g).to(torch.int8)
    bond = torch.triu(bond, diagonal=0)                              # junk the kernel must ignore (upper triangle + diagonal)
    for b in range(B):
        nn = int(n[b])
        el[b, :nn] = torch.tensor([6, 6, 6, 7, 8, 16, 15, 9, 17, 35])[torch.randint(0, 10 if b % 3 == 0 else 5, (nn,), generator=g)]
        bond[b, nn:, :] = torch.randint(0, 5, (D - nn, D), generator=g).to(torch.int8)      # junk rows beyond the molecule
        for i in range(1, nn):
            if b % 4 != 1 or i != nn // 2:                       # every 4th molecule gets a broken chain
                bond[b, i, i - 1] = 4 if (b % 8 == 7) else 1
        if b % 4 == 2 and nn > 2:                                # over-valent: a triple + double bond on one atom
            bond[b, 1, 0] = 3
            bond[b, 2, 1] = 2
        if b % 4 == 3 and nn > 6:                                # ring closure
            bond[b, 5, 0] = 1
    sym0, ok0 = HO.bond_writeback(bond, el, n)
    sym1, ok1 = bond_writeback_hip(bond.to(DEV), el.to(DEV), n.to(DEV))
    assert torch.equal(sym1.cpu(), sym0)
    assert ok1.cpu().tolist() == ok0.tolist()
    assert 10 < int(ok0.sum()) < B - 10                              # the cases above really split both ways
    mols = assemble_molecules(torch.randn(B, D, 3, device=DEV), el.to(DEV), bond.to(DEV), n.to(DEV))
    assert [m.valid for m in mols] == ok0.tolist()
    assert all(m.bond_orders.shape == (int(n[b]), int(n[b])) for b, m in enumerate(mols))
    assert all(torch.equal(m.bond_orders, sym0[b, : int(n[b]), : int(n[b])]) for b, m in enumerate(mols))


def test_ifm_merge_kernel_vs_oracle_and_golden():
    """f3: mcg_ifm_merge (inverse_coord_transform + ifm_prepare_fragments_for_merge, mol_utils.py:460-524, one
    launch) against the reference's own z_known / fixed_mask (golden) and the oracle on a second random case.
    Tolerance: 1e-6 absolute on O(1..10) coordinates (3-term dot products), one-hot / mask channels exact."""
    from ml_conformer_generator_amd import mol_utils as MU
    from oracle import host_oracle as HO
    g = np.load(os.path.join(GOLD, "ifm_front_end.npz"))
    fx = torch.tensor(g["frag_x"])
    fh = MU.one_hot_classes(g["frag_z"].tolist()).float()
    N = int(g["z_known"].shape[1])
    zk, fm = MU.ifm_merge_hip(fx, fh, torch.tensor(g["xg"]), torch.tensor(g["hg"]), torch.tensor(g["shift"]),
                              torch.tensor(g["rotation"]), DEV, N)
    assert torch.equal(fm.cpu(), torch.tensor(g["fixed_mask"]))
    assert float((zk.cpu() - torch.tensor(g["z_known"])).abs().max()) <= 2e-6
    assert torch.equal(zk.cpu()[:, :, 3:], torch.tensor(g["z_known"])[:, :, 3:])
    torch.manual_seed(12)
    B, n_ff, n_gen = 37, 5, 29
    fx2, fh2 = torch.randn(n_ff, 3) * 3, torch.nn.functional.one_hot(torch.randint(0, 8, (n_ff,)), 8).float()
    gx, gh = torch.randn(B, n_gen, 3) * 4, torch.nn.functional.one_hot(torch.randint(0, 8, (B, n_gen)), 8).float()
    rot = torch.linalg.qr(torch.randn(B, 3, 3))[0]
    sh = torch.randn(B, 3)
    zk0, fm0 = HO.ifm_merge_input(fx2, fh2, gx, gh, sh, rot, n_ff + n_gen)
    zk1, fm1 = MU.ifm_merge_hip(fx2, fh2, gx, gh, sh, rot, DEV, n_ff + n_gen)
    assert torch.equal(fm1.cpu(), fm0) and float((zk1.cpu() - zk0).abs().max()) <= 2e-6


def test_config3_ragged_batch_subset_vs_oracle_and_determinism(dyn, edm_sd):
    """BASELINE configs[2] shape: 256 ragged molecules (15..39 atoms, N = 39), full size on the GPU
    (the plan cuts it into three molecule ranges on three streams).  The oracle needs ~1 min per call at this size, so parity is checked on a
    subset of molecules (every sample is independent), plus bit-exact determinism of the whole batch."""
    from oracle import egnn_oracle as EO
    from oracle import host_oracle as HO
    torch.manual_seed(7)
    B, N = 256, 39
    sizes = torch.randint(15, 40, (B,))
    nm, _ = HO.masks_from_sizes(sizes, N)
    z = torch.randn(B, N, 11) * nm
    ctx = torch.tensor([-0.99, -1.66, -1.66]).view(1, 1, 3).repeat(B, N, 1) * nm
    t = torch.full((B, 1), 0.61)
    plan = dyn.plan(sizes, N)
    assert plan.n_edge_tiles >= 5200          # large enough for the plan's automatic split into three ranges
    out1 = dyn.run(plan, t.reshape(-1).to(DEV), z.to(DEV), ctx.to(DEV)).cpu()
    out2 = dyn.run(plan, t.reshape(-1).to(DEV), z.to(DEV), ctx.to(DEV)).cpu()
    assert torch.equal(out1, out2) and torch.isfinite(out1).all()
    pick = [0, 17, 100, 255, int(torch.argmax(sizes)), int(torch.argmin(sizes))]
    nm_s, em_s = HO.masks_from_sizes(sizes[pick], N)
    ref = EO.egnn_dynamics(edm_sd, t[pick], z[pick], nm_s, em_s, ctx[pick])
    ok, err, sc = close(out1[pick], ref)
    assert ok, f"err {err} scale {sc}"


def test_full_sampler_determinism_config2(sampler_factory):
    """Full-size sampler property (64 x 27 atoms, 12 steps): identical device noise seed -> bit-identical
    x, h; different seed -> different samples; masked slots zero; one-hot atom types."""
    gm = sampler_factory(12)
    B, n = 64, 27
    nm = torch.ones(B, n, 1, device=DEV)
    ctx = torch.tensor([-0.99, -1.66, -1.66], device=DEV).view(1, 1, 3).repeat(B, n, 1)
    outs = []
    for seed in (5, 5, 6):
        torch.cuda.manual_seed(seed)
        x, h = gm(nm, None, ctx, 0)
        outs.append((x.cpu(), h.cpu()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert not torch.equal(outs[0][0], outs[2][0])
    assert torch.isfinite(outs[0][0]).all()
    assert float((outs[0][1].sum(2) - 1).abs().max()) == 0.0
    # centre of gravity of every molecule stays at the origin up to fp32 noise of the final noise add
    assert float(outs[0][0].mean(1).abs().max()) < 1e-2 * float(outs[0][0].abs().max())


def test_back_to_back_sampling_runs_do_not_alias(sampler_factory):
    """Two runs over the SAME batch of sizes share the plan-owned latent / context scratch (by design: the captured graph is
    replayed); what the first run returned must not change when the second runs with another context, and the noise
    tensor handed out by `sample_combined_position_feature_noise` is the caller's own."""
    gm = sampler_factory(6)
    B, n = 5, 17
    nm = torch.ones(B, n, 1, device=DEV)
    c1 = torch.tensor([-0.99, -1.66, -1.66], device=DEV).view(1, 1, 3).repeat(B, n, 1)
    c2 = -c1
    torch.cuda.manual_seed(3)
    eps = gm.sample_combined_position_feature_noise(B, n, nm)
    eps_copy = eps.clone()
    x1, h1 = gm(nm, None, c1, 0)
    x1c, h1c = x1.clone(), h1.clone()
    torch.cuda.manual_seed(3)
    gm.sample_combined_position_feature_noise(B, n, nm)
    x2, h2 = gm(nm, None, c2, 0)
    assert torch.equal(x1, x1c) and torch.equal(h1, h1c) and torch.equal(eps, eps_copy)
    assert x1.data_ptr() != x2.data_ptr() and not torch.equal(x1, x2)
    torch.cuda.manual_seed(3)
    gm.sample_combined_position_feature_noise(B, n, nm)
    x3, _ = gm(nm, None, c1, 0)                 # same seed, same context again: the second run left nothing behind
    assert torch.equal(x3, x1)


def test_config2_full_batch_one_call_vs_oracle(dyn, edm_sd):
    """BASELINE configs[1] at FULL size (64 molecules x 27 atoms = 44 928 real edges): one denoiser call of
    the HIP path against the CPU oracle on every element (the oracle needs ~6 s for this on 16 threads)."""
    from oracle import egnn_oracle as EO
    torch.set_num_threads(min(16, torch.get_num_threads()))
    nm, z, ctx, t = _c2_inputs(seed=21)
    B, n, _ = z.shape
    em = edge_mask_of(nm)
    ref = EO.egnn_dynamics(edm_sd, t.reshape(B, 1), z, nm, em, ctx)
    plan = dyn.plan(torch.full((B,), n, dtype=torch.int32), n)
    out = dyn.run(plan, t.to(DEV), z.to(DEV), ctx.to(DEV)).cpu()
    ok, err, sc = close(out, ref)
    assert ok, f"err {err} scale {sc}"
    # the same full-size call in the split-operand mode, same tolerance
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    d6 = EGNNDynamics(device=DEV)
    d6.load_reference_state_dict(edm_sd)
    d6.set_precision("f32x6")
    out6 = d6(t.reshape(B, 1).to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV)).cpu()
    ok, err, sc = close(out6, ref)
    assert ok, f"f32x6 err {err} scale {sc}"


def test_generator_loads_reference_format_checkpoint_files(edm_sd, gcn_sd, tmp_path):
    """`torch.save({"state_dict": ...})` files, as the reference ships them (conformer_generator.py:90-102),
    through the path-based constructor; `__call__` == `generate_conformers` (same positional order)."""
    from ml_conformer_generator_amd import MLConformerGenerator
    e, g = tmp_path / "edm.pt", tmp_path / "seer.pt"
    torch.save({"state_dict": edm_sd}, e)
    torch.save({"state_dict": gcn_sd}, g)
    gen = MLConformerGenerator(diffusion_steps=6, device=DEV, edm_weights=str(e), adj_mat_seer_weights=str(g))
    ctx = torch.tensor([53.6424, 108.3042, 151.4399])
    torch.manual_seed(1); torch.cuda.manual_seed(1)
    a = gen.generate_conformers(None, 3, 1, ctx, 20)
    xa = gen.last_batch["x"].clone()
    torch.manual_seed(1); torch.cuda.manual_seed(1)
    b = gen(None, 3, 1, ctx, 20)
    assert torch.equal(xa, gen.last_batch["x"]) and len(a) == len(b)
    assert gen.last_batch["bond"].dtype == torch.int8 and tuple(gen.last_batch["bond"].shape) == (3, 42, 42)
    bad = dict(edm_sd)
    bad.pop("dynamics.egnn.e_block_4.gcl_1.att_mlp.0.bias")
    with pytest.raises(RuntimeError, match="Missing key"):
        MLConformerGenerator(diffusion_steps=6, device=DEV, edm_weights=bad, adj_mat_seer_weights=gcn_sd)


def test_shape_tanimoto_kernel_vs_golden_and_oracle():
    """f4 (grid half): batched HIP shape-Tanimoto vs the reference's scores (golden) and the oracle.
    Tolerance 2e-4 absolute: the reference's own distances come from a matmul-based fp32 `cdist`
    (|a|^2+|b|^2-2ab), the kernel uses direct differences."""
    from ml_conformer_generator_amd.cheminformatics import shape_tanimoto_batch, tanimoto_score
    from oracle import shape_oracle as SO
    g = load_golden("shape_tanimoto.npz")
    ref = g["xyz_ceyyag"]
    names = ("yibfeu", "ceyyag")
    N = 23
    cands = torch.zeros(2, N, 3)
    n_nodes = torch.tensor([23, 17])
    cands[0, :23] = g["xyz_yibfeu"]
    cands[1, :17] = g["xyz_ceyyag"]
    scores, best, which = shape_tanimoto_batch(ref, cands, n_nodes, device=DEV)
    scores = scores.cpu()
    assert float((scores[0].double() - g["ceyyag__yibfeu"]).abs().max()) < 2e-4
    assert float((scores[1].double() - g["ceyyag__ceyyag"]).abs().max()) < 2e-4
    assert int(which[1]) == 0 and abs(float(best[1]) - 1.0) < 1e-5       # a molecule matches itself, unrotated
    assert int(which[0]) == int(torch.argmax(g["ceyyag__yibfeu"]))
    # larger batch vs the oracle's orientation search
    torch.manual_seed(0)
    B = 6
    many = torch.randn(B, 27, 3) * 2.0
    nn = torch.tensor([27, 20, 15, 27, 22, 18])
    sc, bst, wh = shape_tanimoto_batch(ref, many, nn, device=DEV)
    for b in range(B):
        ob, ow = SO.best_orientation_score(ref, many[b, : int(nn[b])])
        assert abs(float(bst[b]) - ob) < 2e-4
    assert abs(tanimoto_score(g["xyz_yibfeu"], g["xyz_paba"]) - float(g["yibfeu__paba"][0])) < 2e-4


def test_integration_md_ctypes_stub_runs_as_documented(edm_sd, gcn_sd):
    """Executes the reference-side binding printed in INTEGRATION.md section 3 verbatim (only the library
    path is made absolute) around stand-ins that expose what the reference's modules expose
    (`.egnn.state_dict()`, `.state_dict()`), with FRESH tensors on every call as the reference's sampler loop
    produces them - which also drives the plan's staged-graph mode - and checks both seams against the oracle."""
    import re
    from ml_conformer_generator_amd import _lib
    from oracle import egnn_oracle as EO
    from oracle import gcn_oracle as GO
    from oracle import host_oracle as HO
    from ml_conformer_generator_amd.synthetic import synth_gcn_inputs
    import os
    md = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    stub = next(b for b in blocks if "class HipEGNNDynamics" in b)
    stub = stub.replace('C.CDLL("libmlconfgen_hip.so")', f'C.CDLL({_lib.LIB_PATH!r})')
    ns = {}
    exec(compile(stub, "INTEGRATION.md#3", "exec"), ns)

    class _Egnn:
        def state_dict(self):
            return {k[len("dynamics.egnn."):]: v for k, v in edm_sd.items() if k.startswith("dynamics.egnn.")}

    class _RefDynamics:
        egnn = _Egnn()

    dyn = ns["HipEGNNDynamics"](_RefDynamics())
    sizes = torch.tensor([17, 15, 19, 16])
    N = 19
    nm, em = HO.masks_from_sizes(sizes, N)
    ctx = torch.tensor([-0.9, -1.6, -1.7]).view(1, 1, 3).repeat(4, N, 1) * nm
    for it in range(4):                     # new xh / t / out tensors every call
        torch.manual_seed(100 + it)
        xh = torch.randn(4, N, 11) * nm
        t = torch.full((4, 1), 0.1 + 0.2 * it)
        keep = [torch.empty(1000 + 64 * it, device=DEV)]       # shifts the caching allocator's addresses
        out = dyn(t.to(DEV), xh.to(DEV), nm.to(DEV), em.to(DEV), ctx.clone().to(DEV)).cpu()
        ref = EO.egnn_dynamics(edm_sd, t, xh, nm, em, ctx)
        ok, err, sc = close(out, ref)
        assert ok, (it, err, sc)
        del keep
    # the stub checks a pair of mask tensors ONCE (the reference's loop passes the same two every step) and refuses
    # anything but the canonical masks - also an accepted mask that was edited in place
    nm_d, em_d = nm.to(DEV), em.to(DEV)
    dyn(t.to(DEV), xh.to(DEV), nm_d, em_d, ctx.to(DEV))
    checked = dyn._masks
    dyn(t.to(DEV), xh.to(DEV), nm_d, em_d, ctx.to(DEV))
    assert dyn._masks is checked
    bad = em_d.clone()
    bad[int(torch.nonzero(bad)[5, 0])] = 0.0
    with pytest.raises(ValueError, match="canonical"):
        dyn(t.to(DEV), xh.to(DEV), nm_d, bad, ctx.to(DEV))
    em_d[int(torch.nonzero(em_d)[0, 0])] = 0.0
    with pytest.raises(ValueError, match="canonical"):
        dyn(t.to(DEV), xh.to(DEV), nm_d, em_d, ctx.to(DEV))

    class _RefSeer:
        device, dimension = DEV, 42

        def state_dict(self):
            return gcn_sd

    seer = ns["HipAdjMatSeer"](_RefSeer())
    el, dm, am = synth_gcn_inputs(3, [17, 27, 39], seed=5)
    logits = seer(el.to(DEV), dm.to(DEV), am.to(DEV)).cpu()
    ref = GO.adj_mat_seer(gcn_sd, el, dm, am)
    ok, err, sc = close(logits, ref)
    assert ok, (err, sc)
    assert torch.equal(logits.argmax(-1), ref.argmax(-1))


def test_handles_are_freed(edm_sd):
    """Model / plan handles release their device memory (weights ~200 MB per model incl. bf16 packs, plan
    buffers, HIP graph, streams): repeated create/destroy must not grow the footprint."""
    import gc
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    sizes = torch.randint(15, 40, (32,))
    nm = (torch.arange(39).unsqueeze(0) < sizes.unsqueeze(1)).float().unsqueeze(2)
    z = (torch.randn(32, 39, 11) * nm).to(DEV)
    ctx = (torch.randn(32, 1, 3).repeat(1, 39, 1) * nm).to(DEV)
    t = torch.full((32, 1), 0.3, device=DEV)

    def cycle():
        d = EGNNDynamics(device=DEV)
        d.load_reference_state_dict(edm_sd)
        for prec in ("f32", "bf16"):
            d.set_precision(prec)
            d(t, z, nm.to(DEV), None, ctx)
        torch.cuda.synchronize()
        del d
        gc.collect()

    cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(4):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, (free0, free1)


def test_evaluate_shape_vs_reference_pipeline():
    """Shape half of evaluate_samples (pipeline.py:30-85) on the device: principal frames + four-orientation
    HIP overlap search against the reference's frames / best scores (golden)."""
    from ml_conformer_generator_amd.cheminformatics import evaluate_shape, shape_quadrupole_batch
    g = load_golden("shape_quadrupole.npz")
    names = ["yibfeu", "walk27", "paba", "ceyyag"]
    N = 27
    coords, nn = torch.zeros(len(names), N, 3), []
    for i, n in enumerate(names):
        x = g[f"xyz_{n}"]
        coords[i, : x.shape[0]] = x + torch.tensor([0.7, -1.1, 0.4])       # evaluate_shape re-centres
        nn.append(x.shape[0])
    nn = torch.tensor(nn)
    mom, frames = shape_quadrupole_batch(coords - torch.tensor([0.7, -1.1, 0.4]) * (torch.arange(N).view(1, N, 1) < nn.view(-1, 1, 1)),
                                         nn, device=DEV)
    for i, n in enumerate(names):
        assert float((mom[i].cpu() - g[f"moments_{n}"]).abs().max()) < 2e-5, n
        assert float((frames[i, : int(nn[i])].cpu() - g[f"frame_{n}"]).abs().max()) < 2e-5, n
    ref_pf, res = evaluate_shape(g["xyz_ceyyag"], coords, nn, device=DEV)
    assert float((ref_pf - g["frame_ceyyag"]).abs().max()) < 2e-5
    for i, key in ((0, "best__ceyyag__yibfeu"), (1, "best__ceyyag__walk27")):
        assert abs(res[i]["shape_tanimoto"] - float(g[key][0])) < 2e-4, key
        assert res[i]["orientation"] == int(g[key][1]), key
    assert res[3]["orientation"] == 0 and abs(res[3]["shape_tanimoto"] - 1.0) < 1e-4      # the reference itself
    assert res[2]["coords"].shape == (10, 3)


def test_f32x6_mode_matches_fp32_tolerance(edm_sd):
    """"f32x6": the edge MLP's 420x420 contraction as six bf16 partial products of three-part operands, fp32
    accumulate.  It must meet the SAME stated fp32 tolerance against the oracle as the exact-fp32 path, on a
    ragged batch and on the golden seam vectors, and be deterministic."""
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    from oracle import egnn_oracle as EO
    from oracle import host_oracle as HO
    d = EGNNDynamics(device=DEV)
    d.load_reference_state_dict(edm_sd)
    d.set_precision("f32x6")
    torch.manual_seed(21)
    sizes = torch.tensor([19, 27, 15, 33, 39, 22])
    N = 39
    nm, em = HO.masks_from_sizes(sizes, N)
    z = torch.randn(6, N, 11) * nm
    ctx = torch.randn(6, 1, 3).repeat(1, N, 1) * nm
    t = torch.full((6, 1), 0.35)
    out = d(t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV)).cpu()
    assert next(iter(d._plans.values())).edge_mt == 4
    ref = EO.egnn_dynamics(edm_sd, t, z, nm, em, ctx)
    ok, err, sc = close(out, ref)
    assert ok, (err, sc)
    out2 = d(t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV)).cpu()
    assert torch.equal(out, out2)
    assert float((out * (1 - nm)).abs().max()) == 0.0
    for tag in ("b2n20", "b3n39"):
        g = load_golden(f"dynamics_{tag}.npz")
        o = d(g["t"].to(DEV), g["xh"].to(DEV), g["node_mask"].to(DEV), edge_mask_of(g["node_mask"]).to(DEV),
              g["context"].to(DEV)).cpu()
        ok, err, sc = close(o, g["out"])
        assert ok, (tag, err, sc)
    d.set_precision("f32")
    out32 = d(t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV)).cpu()
    e6 = float((out - ref).abs().max()) / float(ref.abs().max())
    e32 = float((out32 - ref).abs().max()) / float(ref.abs().max())
    print(f"max rel err vs oracle: f32x6 {e6:.2e}, exact fp32 {e32:.2e}")
    assert e6 < 5e-6


def test_split_operand_modes_are_as_accurate_as_fp32_against_fp64(edm_sd):
    """Ground truth = the oracle evaluated in fp64 (weights and inputs are exactly representable, so this is the
    real-number value of the network up to 1e-16).  The exact-fp32 kernel and f32x6 must sit at the same
    distance from it as the fp32 CPU evaluation does - i.e. the split-operand modes lose nothing measurable - while
    the bf16 mode is two orders of magnitude further away."""
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    from oracle import egnn_oracle as EO
    from oracle import host_oracle as HO
    sd64 = {k: v.double() for k, v in edm_sd.items()}
    torch.manual_seed(21)
    sizes = torch.randint(15, 40, (12,))
    N = 39
    nm, em = HO.masks_from_sizes(sizes, N)
    z = torch.randn(12, N, 11) * nm
    ctx = torch.randn(12, 1, 3).repeat(1, N, 1) * nm
    t = torch.full((12, 1), 0.35)
    ref64 = EO.egnn_dynamics(sd64, t.double(), z.double(), nm.double(), em.double(), ctx.double())
    ref32 = EO.egnn_dynamics(edm_sd, t, z, nm, em, ctx)
    sc = float(ref64.abs().max())

    def rms(o):
        return float((o.double() - ref64).pow(2).mean().sqrt()) / sc

    base = rms(ref32)
    d = EGNNDynamics(device=DEV)
    d.load_reference_state_dict(edm_sd)
    got = {}
    for mode in ("f32", "f32x6", "bf16"):
        d.set_precision(mode)
        got[mode] = rms(d(t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV)).cpu())
    print("rms deviation from fp64 / max|out|: fp32 CPU %.2e, " % base + ", ".join(f"{k} {v:.2e}" for k, v in got.items()))
    assert got["f32"] < 1.5 * base and got["f32x6"] < 1.5 * base
    assert abs(got["f32x6"] - got["f32"]) < 0.2 * got["f32"]
    assert got["bf16"] > 20 * got["f32"]


def test_split_operand_modes_under_cancellation_and_wide_dynamic_range_vs_fp64(edm_sd):
    """Adversarial numerics for the split-operand contraction, ground truth = the oracle in fp64.
    * Cancellation: every hidden layer is rebuilt so that channel k + 210 of its first Linear duplicates channel k, the
      following Linear weighs the pair (w, -(1 + 2^-7) w) and has no bias: each 420-term dot product is then 128x smaller
      than its positive and negative halves, and nothing else is added to it - what survives the cancellation is exactly
      what a contraction that dropped low operand bits would get wrong.  The fp32 evaluations themselves lose 2-4
      digits of the velocity here (1e-5 .. 1e-3 of its magnitude against fp64).
    * Dynamic range: per-molecule feature scales from 1e-3 to 1e3, coordinate scales from 0.1 to 30.
    f32x6 must sit at the same distance from fp64 as the exact fp32 kernel and the fp32 CPU evaluation do, per
    molecule (each against its own magnitude) and separately for the velocity and the feature channels."""
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    from oracle import egnn_oracle as EO
    from oracle import host_oracle as HO
    delta = 2.0 ** -7
    sd = {k: v.clone() for k, v in edm_sd.items()}
    for k in list(sd):
        for first, second in (("edge_mlp.0.", "edge_mlp.2."), ("coord_mlp.0.", "coord_mlp.2."), ("node_mlp.0.", "node_mlp.2.")):
            if k.endswith(first + "weight"):
                base = k[: -len(first + "weight")]
                sd[k][210:] = sd[k][:210]
                sd[base + first + "bias"][210:] = sd[base + first + "bias"][:210]
                w2 = sd[base + second + "weight"]
                w2[:, 210:] = -(1.0 + delta) * w2[:, :210]
                sd[base + second + "bias"].zero_()
    sd64 = {k: v.double() for k, v in sd.items()}
    torch.manual_seed(33)
    B, N = 10, 30
    sizes = torch.randint(12, 31, (B,))
    nm, em = HO.masks_from_sizes(sizes, N)
    z = torch.randn(B, N, 11) * nm
    z[:, :, 3:] *= (10.0 ** torch.linspace(-3, 3, B)).view(B, 1, 1)          # features: six decades over the batch
    z[:, :, :3] *= (10.0 ** torch.linspace(-1, 1.5, B)).view(B, 1, 1)        # coordinates: 0.1 .. 30
    ctx = (torch.randn(B, 1, 3) * (10.0 ** torch.linspace(2, -2, B)).view(B, 1, 1)).repeat(1, N, 1) * nm
    t = torch.rand(B, 1)
    ref64 = EO.egnn_dynamics(sd64, t.double(), z.double(), nm.double(), em.double(), ctx.double())
    ref32 = EO.egnn_dynamics(sd, t, z, nm, em, ctx)
    assert bool(torch.isfinite(ref64).all())

    def dev(o, sl):                    # per molecule, against ITS OWN magnitude: the scales differ by decades
        e = (o.double() - ref64)[:, :, sl].flatten(1).pow(2).mean(1).sqrt()
        return e / ref64[:, :, sl].flatten(1).abs().amax(1).clamp_min(1e-30)

    groups = {"velocity": slice(0, 3), "features": slice(3, 11)}
    base = {g: dev(ref32, sl) for g, sl in groups.items()}
    assert float(base["velocity"].median()) > 3e-6     # the cancellation is visible: fp32 itself loses digits here
    d = EGNNDynamics(device=DEV)
    d.load_reference_state_dict(sd)
    for mode in ("f32", "f32x6"):
        d.set_precision(mode)
        out = d(t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV)).cpu()
        assert bool(torch.isfinite(out).all())
        for g, sl in groups.items():
            v = dev(out, sl)
            print(f"{mode:6s} {g:9s} deviation from fp64, median / max over molecules: {float(v.median()):.2e} / {float(v.max()):.2e}"
                  f"   (fp32 CPU: {float(base[g].median()):.2e} / {float(base[g].max()):.2e})")
            # rounding noise of a different summation order, not a lost operand bit: within 4x per molecule, 2x in the median
            assert bool((v <= 4.0 * base[g] + 1e-7).all()), (mode, g, v.tolist(), base[g].tolist())
            assert float(v.median()) <= 2.0 * float(base[g].median()) + 1e-7, (mode, g)


def test_sampler_trajectory_in_f32x6_mode_vs_golden(edm_sd):
    """The reference sampler trajectory (golden, recorded noise tape) reproduced with the split-operand kernels:
    same tolerance as the exact-fp32 path (rel 1e-3 of max|z| per step, atom types exact)."""
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    from ml_conformer_generator_amd.equivariant_diffusion import EquivariantDiffusion, PredefinedNoiseSchedule
    g = load_golden("sampler_T20_b4n19.npz")
    nm = g["node_mask"]
    d = EGNNDynamics(device=DEV)
    d.load_reference_state_dict(edm_sd)
    d.set_precision("f32x6")
    T = int(g["T"])
    gm = EquivariantDiffusion(dynamics=d, in_node_nf=8, timesteps=1000, noise_precision=1e-5)
    gm.gamma = PredefinedNoiseSchedule(timesteps=T, precision=1e-5)
    gm.T = T
    gm.noise_fn = TapeNoise(g["noise"], DEV)
    gm.trace = []
    x, h = gm(nm.to(DEV), edge_mask_of(nm).to(DEV), g["context"].to(DEV), 0)
    zt = torch.stack(gm.trace).cpu()
    ref = g["z_trace"]
    rel = float((zt - ref).abs().max()) / float(ref.abs().max())
    assert rel < 1e-3, rel
    assert torch.equal(h.cpu().argmax(2) * nm.squeeze(2).long(), g["h"].argmax(2) * nm.squeeze(2).long())
    assert float((x.cpu() - g["x"]).abs().max()) / float(g["x"].abs().max()) < 1e-3


@pytest.mark.parametrize("mode", ["f32", "f32x6", "bf16"])
def test_randomized_batch_shapes_vs_oracle(edm_sd, mode):
    """Random batch compositions (1..24 molecules of 2..42 atoms, pad width up to 42, both plan families, tiles
    that straddle many small molecules) against the oracle: fp32 tolerance for the exact and split-operand modes,
    the stated bf16 tolerance (3e-2 of max|out|) for the bf16 mode."""
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    from oracle import egnn_oracle as EO
    from oracle import host_oracle as HO
    d = EGNNDynamics(device=DEV)
    d.load_reference_state_dict(edm_sd)
    d.set_precision(mode)
    g = torch.Generator().manual_seed(2024)
    for trial in range(10):
        B = int(torch.randint(1, 25, (1,), generator=g))
        lo = [2, 6, 15][trial % 3]
        hi = [9, 42, 39][trial % 3]
        sizes = torch.randint(lo, hi + 1, (B,), generator=g)
        N = int(sizes.max()) + int(torch.randint(0, 3, (1,), generator=g))
        N = min(N, 42)
        nm, em = HO.masks_from_sizes(sizes, N)
        z = torch.randn(B, N, 11, generator=g) * nm
        ctx = torch.randn(B, 1, 3, generator=g).repeat(1, N, 1) * nm
        t = torch.rand(B, 1, generator=g)
        ref = EO.egnn_dynamics(edm_sd, t, z, nm, em, ctx)
        out = d(t.to(DEV), z.to(DEV), nm.to(DEV), em.to(DEV), ctx.to(DEV)).cpu()
        assert float((out * (1 - nm)).abs().max()) == 0.0
        if mode == "bf16":
            assert float((out - ref).abs().max()) <= 3e-2 * max(1.0, float(ref.abs().max())), (trial, B, sizes.tolist())
        else:
            ok, err, sc = close(out, ref)
            assert ok, (trial, B, sizes.tolist(), err, sc)


def test_f32x6_end_to_end_equals_exact_path_on_a_contractive_network(gcn_sd):
    """The whole generation (seeded size draw, T = 60 sampler, hand-off, GCN, bond argmax) in f32x6 and in exact fp32 from
    the same seeds, on synthetic weights under which the sampler is contractive (nn.Linear-family init x 0.3): atom
    types and the adjacency (bond-order argmax of the real lower triangle) must be IDENTICAL for every molecule and the
    coordinates within 1e-5 of max|x|.  (On the expansive "v2d" weights the untrained sampler is chaotic - two exact-fp32
    runs that differ by one ulp of the context end as far apart as f32x6 and fp32 do: tools/x6_end_to_end.py.)"""
    from ml_conformer_generator_amd import MLConformerGenerator, weights as W
    from ml_conformer_generator_amd.synthetic import DUMMY_CONTEXT
    sd = W.synth_edm_state_dict(1234, weight_gain=0.3)
    ctx = torch.tensor(DUMMY_CONTEXT)
    res = {}
    for mode in ("f32", "f32x6"):
        gen = MLConformerGenerator(diffusion_steps=60, device=DEV, edm_weights=sd, adj_mat_seer_weights=gcn_sd, compute_dtype=mode)
        for name, (n_samples, variance) in {"c1": (32, 0), "c2": (48, 12)}.items():
            torch.default_generator.manual_seed(7)
            torch.cuda.manual_seed(7)
            gen._generate_shard(ctx, 27, variance, None, n_samples, 0, None, True, 3, 50)
            res[(mode, name)] = {k: v.cpu() for k, v in gen.last_batch.items()}
    for name in ("c1", "c2"):
        a, b = res[("f32", name)], res[("f32x6", name)]
        assert torch.equal(a["n_nodes"], b["n_nodes"])
        assert bool(torch.isfinite(a["x"]).all()) and bool(torch.isfinite(b["x"]).all())
        assert float((a["x"] - b["x"]).abs().max()) <= 1e-5 * float(a["x"].abs().max())
        assert torch.equal(a["h"].argmax(2), b["h"].argmax(2))
        assert torch.equal(a["elements"], b["elements"])
        n = a["n_nodes"]
        D = a["bond"].shape[1]
        inside = (torch.arange(D).view(1, D, 1) < n.view(-1, 1, 1)) & (torch.arange(D).view(1, 1, D) < n.view(-1, 1, 1))
        m = torch.tril(torch.ones(D, D, dtype=torch.bool), -1).unsqueeze(0) & inside
        assert int(((a["bond"] != b["bond"]) & m).sum()) == 0


@pytest.mark.parametrize("n_samples,variance", [(64, 0), (256, 12)])
def test_bench_workload_stays_finite_over_100_steps(gcn_sd, n_samples, variance):
    """The weights bench.py times (recipe "v2d") keep the untrained T = 100 sampler finite at configs[1] and at the
    configs[2] shape - every latent of the trajectory and every output coordinate - so the benchmark does not time
    NaN arithmetic.  (The parity fixtures' recipe "v2" overflows at this size from step ~30: asserted too, so that a
    change of recipe that silently fixes or breaks either side is noticed.)"""
    from ml_conformer_generator_amd import MLConformerGenerator, weights as W
    from ml_conformer_generator_amd.synthetic import DUMMY_CONTEXT
    ctx = torch.tensor(DUMMY_CONTEXT)
    gen = MLConformerGenerator(diffusion_steps=100, device=DEV, edm_weights=W.synth_edm_state_dict(1234, recipe="v2d"),
                               adj_mat_seer_weights=gcn_sd)
    gm = gen.generative_model
    gm.trace = []
    torch.default_generator.manual_seed(7)
    torch.cuda.manual_seed(7)
    gen._generate_shard(ctx, 27, variance, None, n_samples, 0, None, True, 3, 50)
    assert len(gm.trace) == 100
    assert all(bool(torch.isfinite(z).all()) for z in gm.trace)
    assert max(float(z.abs().max()) for z in gm.trace) < 1e5
    assert bool(torch.isfinite(gen.last_batch["x"]).all())
    gm.trace = None
    if n_samples == 64:
        bad = MLConformerGenerator(diffusion_steps=100, device=DEV, edm_weights=W.synth_edm_state_dict(1234),
                                   adj_mat_seer_weights=gcn_sd)
        torch.default_generator.manual_seed(7)
        torch.cuda.manual_seed(7)
        bad._generate_shard(ctx, 27, 0, None, 64, 0, None, True, 3, 50)
        assert not bool(torch.isfinite(bad.last_batch["x"]).all())


def test_sharded_path_on_one_rank_rccl_in_a_fresh_process():
    """Multi-GPU readiness (no 8-GPU box is available to the build): the RCCL code path of the sharded product entry point
    on HARDWARE.  A fresh child process (never a re-exec of one that touched the GPU) initialises a 1-rank `nccl` group
    with `device_id`, forces the collectives (MCG_FORCE_COLLECTIVE=1: status byte + result all-gather / gather on device
    tensors) and checks `generate_conformers_sharded` against the unsharded call under the same seeds - bit-identical
    tensors.  tools/rccl_one_rank_check.py prints one JSON line."""
    import json
    import subprocess
    import sys
    from conftest import REPO
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "rccl_one_rank_check.py")], capture_output=True,
                       text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    print(line)
    assert line["backend"] == "nccl" and line["world"] == 1 and line["identical_to_unsharded"] and line["finite"]
    assert len(line["sizes"]) == 8 and 17 <= min(line["sizes"]) and max(line["sizes"]) <= 23


def test_new_size_vector_every_call_reuses_pooled_plan_memory(edm_sd):
    """A ragged caller meets a never-seen size vector on every call (reference protocol: 1 000 different references,
    research_scripts/evaluation.py:98-103): with a bounded LRU plan cache every call builds a plan and evicts one.  The
    plans' device blocks come from the library's pool (csrc/mcg_devmem.hip), so after the first few calls the driver is not
    asked for memory any more and free device memory stays flat; a plan built on RECYCLED blocks computes bit-identical
    results to one built on fresh driver memory."""
    import numpy as np
    from ml_conformer_generator_amd import _lib
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    L = _lib.lib()

    def stats(trim=0):
        st = np.zeros(4, dtype=np.int64)
        _lib.check(L.mcg_pool_stats(st.ctypes.data, trim), "mcg_pool_stats")
        return [int(v) for v in st]

    d = EGNNDynamics(device=DEV)
    d.load_reference_state_dict(edm_sd)
    d.plan_cache_size = 3
    g = torch.Generator().manual_seed(123)
    B, N = 96, 39
    free_mem, allocs, outs, inputs = [], [], [], []
    for k in range(18):
        sizes = torch.randint(15, 40, (B,), generator=g)
        nm = (torch.arange(N).unsqueeze(0) < sizes.unsqueeze(1)).float().unsqueeze(2).to(DEV)
        xh = torch.randn(B, N, 11, generator=g).to(DEV) * nm
        cx = torch.randn(B, 1, 3, generator=g).repeat(1, N, 1).to(DEV) * nm
        t = torch.full((B,), 0.4, device=DEV)
        plan = d.plan(sizes, N)
        out = d.run(plan, t, xh, cx)
        torch.cuda.synchronize(DEV)
        assert bool(torch.isfinite(out).all())
        free_mem.append(torch.cuda.mem_get_info(DEV)[0])
        allocs.append(stats()[2])
        if k >= 16:
            outs.append(out.clone()); inputs.append((sizes, t, xh, cx))
        del plan
    assert len(d._plans) == 3
    assert allocs[-1] - allocs[9] <= 2, allocs                      # steady state: (almost) no driver allocation per new plan
    assert stats()[3] > 20                                          # ... because the pool serves them
    assert free_mem[9] - free_mem[-1] <= 96 << 20, free_mem         # flat device memory (a stray new size class at most)
    # LRU, not FIFO: a plan that keeps being used survives newer ones
    keep_sizes, t, xh, cx = inputs[0]
    key_plan = d.plan(keep_sizes, N)
    for k in range(4):
        d.plan(torch.randint(15, 40, (B,), generator=g), N)
        assert d.plan(keep_sizes, N) is key_plan
    # recycled blocks vs fresh driver memory: bit-identical
    d._plans.clear()
    import gc
    gc.collect()
    stats(trim=1)
    assert stats()[1] == 0
    for (sizes, t, xh, cx), ref in zip(inputs, outs):
        assert torch.equal(d.run(d.plan(sizes, N), t, xh, cx), ref)


def test_bf16_mode_gathers_partial_sums_in_the_node_gemm(edm_sd):
    """configs[4] arithmetic at (half) its per-GPU size: in bf16 mode the node GEMM reads an atom's aggregate straight from
    the 64-row edge kernel's per-unit partial sums (<= 4 rows of P summed and divided by 100 in its A-loader - no combine
    launch) and carries the coordinate update as a side job (no update launch).  Same operand rounding and the same k order
    per output element at every wave-tile width, so
      * the call must be BIT-IDENTICAL between the wave-tile widths (MCG_OPT_GEMM_RN; the launch counters say which ran), on
        one and on two molecule ranges;
      * ONE GCL layer must be bit-identical between the gathering GEMM and the stand-alone combine route
        (`mcg_egnn_gcl_debug` materialises the aggregate);
      * the whole call stays within the bf16 tolerance of the fp32 path."""
    import numpy as np
    from ml_conformer_generator_amd import _lib
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    L = _lib.lib()

    def launches():
        c = np.zeros(32, dtype=np.int64)
        _lib.check(L.mcg_debug_gemm_launches(c.ctypes.data, 1), "mcg_debug_gemm_launches")
        return c.reshape(4, 8)

    g = torch.Generator().manual_seed(77)
    B, N = 160, 39
    sizes = torch.randint(15, 40, (B,), generator=g)
    assert int(sizes.sum()) >= 2 * 2048
    nm = (torch.arange(N).unsqueeze(0) < sizes.unsqueeze(1)).float().unsqueeze(2)
    z = (torch.randn(B, N, 11, generator=g) * nm).to(DEV)
    ctx = (torch.randn(B, 1, 3, generator=g).repeat(1, N, 1) * nm).to(DEV)
    t = torch.full((B,), 0.3, device=DEV)
    d = EGNNDynamics(device=DEV)
    d.load_reference_state_dict(edm_sd)
    ref32 = d.run(d.plan(sizes, N), t, z, ctx).clone()
    d.set_precision("bf16")
    outs = []
    for n_ranges in (1, 2):
        plan = d.plan(sizes, N, n_ranges=n_ranges)
        assert plan.edge_mt == 4
        _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_NODE_FUSED, 1), "mcg_egnn_set_option")       # three launches per layer (round 5)
        _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_GEMM_BF16_LDS, 1), "mcg_egnn_set_option")    # the 32-row kernel
        for rn in (0, 1, 2, 3):
            _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_GEMM_RN, rn), "mcg_egnn_set_option")
            launches()
            outs.append(d.run(plan, t, z, ctx).clone())
            c = launches()
            assert int(c[1].sum()) == 63 * n_ranges and int(c[1][7]) == 0, c
            if rn:
                assert int(c[1][rn]) == 63 * n_ranges, (rn, c)
        _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_GEMM_RN, 0), "mcg_egnn_set_option")
        # round 5: the LDS-staged 9-wave kernel (activation block parked once per workgroup) - forced, then the automatic choice
        _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_GEMM_BF16_LDS, 2), "mcg_egnn_set_option")
        launches()
        outs.append(d.run(plan, t, z, ctx).clone())
        c = launches()
        assert int(c[1][7]) == 63 * n_ranges and int(c[1].sum()) == 63 * n_ranges, c
        _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_GEMM_BF16_LDS, 0), "mcg_egnn_set_option")
        launches()
        outs.append(d.run(plan, t, z, ctx).clone())
        c = launches()
        assert int(c[1].sum()) == 63 * n_ranges, c
        if n_ranges == 1:
            assert int(c[1][7]) == 63, c                  # ~4 400 atoms in one range: the automatic choice is the LDS kernel
        # round 6: the node phase of a GCL layer as ONE launch (W3 -> SiLU -> W4 -> + h -> the next edge layer's first-layer
        # projections, mcg_node_fused.h): forced, then the automatic choice - 18 fused launches + one first-layer GEMM per block
        # (it carries the coordinate update) instead of 63 GEMM launches, and not a bit of difference
        for fused in (2, 0):
            _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_NODE_FUSED, fused), "mcg_egnn_set_option")
            launches()
            outs.append(d.run(plan, t, z, ctx).clone())
            c = launches()
            if fused == 2 or n_ranges == 1:
                assert int(c[1][6]) == 18 * n_ranges and int(c[1].sum()) == 27 * n_ranges, (fused, n_ranges, c)
            else:
                assert int(c[1].sum()) in (27 * n_ranges, 63 * n_ranges), c
        _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_NODE_FUSED, 1), "mcg_egnn_set_option")
    for k, o in enumerate(outs):
        assert torch.equal(o, outs[8 * (k // 8)]), k        # kernels, tile widths and the launch structure do not change a bit
        # (molecule ranges cut the 64-row units elsewhere: an atom's partial sums split differently - fp32 re-association
        #  that the bf16 operand rounding can amplify to a rounding flip)
        assert float((o - outs[0]).abs().max()) <= 3e-3 * float(ref32.abs().max())
        assert float((o - ref32).abs().max()) <= 3e-2 * float(ref32.abs().max())
    # one GCL layer: gathered in the GEMM (block hook) vs materialised by the combine kernel (GCL hook), same plan -
    # with the 32-row kernel and with the LDS-staged one
    plan = d.plan(sizes, N, n_ranges=1)
    M = plan.n_real_nodes
    h0 = torch.randn(M, 420, generator=g).to(DEV)
    x0 = (torch.randn(M, 3, generator=g) * 2).to(DEV)
    blocks = []
    for lds, fused in ((1, 1), (2, 1), (2, 2)):
        _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_GEMM_BF16_LDS, lds), "mcg_egnn_set_option")
        _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_NODE_FUSED, fused), "mcg_egnn_set_option")
        via_combine = d.gcl_debug(plan, 0, h0, x0, x0)["h_out"]
        launches()
        h_blk, x_blk = d.block_debug(plan, 0, h0, x0, x0)         # gcl_0, gcl_1, coordinate layer
        c = launches()
        assert int(c[1][6]) == (2 if fused == 2 else 0) and int(c[1].sum()) == (3 if fused == 2 else 7), (lds, fused, c)
        via2 = d.gcl_debug(plan, 1, via_combine, x0, x0)["h_out"]
        assert torch.equal(h_blk, via2), (lds, fused)
        blocks.append((h_blk.clone(), x_blk.clone()))
    assert all(torch.equal(b[0], blocks[0][0]) and torch.equal(b[1], blocks[0][1]) for b in blocks)     # x too: the fused launch's Pab feeds the coordinate layer
    _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_GEMM_BF16_LDS, 0), "mcg_egnn_set_option")
    _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_NODE_FUSED, 1), "mcg_egnn_set_option")
    # the LDS-staged kernel at shapes the automatic choice never gives it: a ragged last row block (M % 32 = 1, 31), fewer rows
    # than one block, a single molecule - forced on against forced off, bit for bit
    for small in ([17, 16], [31], [6, 7, 9, 11], [39] * 3 + [22]):
        sz = torch.tensor(small)
        Ns = int(sz.max())
        nms = (torch.arange(Ns).unsqueeze(0) < sz.unsqueeze(1)).float().unsqueeze(2)
        zs = (torch.randn(len(small), Ns, 11, generator=g) * nms).to(DEV)
        cs = (torch.randn(len(small), 1, 3, generator=g).repeat(1, Ns, 1) * nms).to(DEV)
        ts = torch.full((len(small),), 0.6, device=DEV)
        res = []
        for lds in (1, 2):
            _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_GEMM_BF16_LDS, lds), "mcg_egnn_set_option")
            launches()
            res.append(d.run(d.plan(sz, Ns), ts, zs, cs).clone())
            c = launches()
            assert (int(c[1][7]) > 0) == (lds == 2), (small, lds, c)
        assert torch.equal(res[0], res[1]), small
        # ... and the fused node launch at the same shapes (ragged last row block, fewer rows than one block)
        _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_GEMM_BF16_LDS, 0), "mcg_egnn_set_option")
        _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_NODE_FUSED, 2), "mcg_egnn_set_option")
        launches()
        res.append(d.run(d.plan(sz, Ns), ts, zs, cs).clone())
        c = launches()
        assert int(c[1][6]) == 18, (small, c)
        assert torch.equal(res[0], res[2]), small
        _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_NODE_FUSED, 1), "mcg_egnn_set_option")
    _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_GEMM_BF16_LDS, 0), "mcg_egnn_set_option")
    _lib.check(L.mcg_egnn_set_option(d.handle, _lib.OPT_NODE_FUSED, 0), "mcg_egnn_set_option")


@pytest.mark.gpu
def test_plans_destroyed_on_another_thread_while_this_one_captures_graphs(edm_sd):
    """Round-5 review, weak #13: `BatchPlan.__del__` may run on any thread (the garbage collector's, a worker's) while the calling
    thread is inside a HIP-graph capture of another plan.  The library captures in THREAD-LOCAL mode and `mcg_plan_destroy` waits on
    the plan's own event on the plan's own device, so this must neither break the capture nor the results: one thread creates, runs
    and destroys plans of ever-new shapes as fast as it can while the main thread runs first calls (plan + capture) over fresh size
    vectors and compares every result with an undisturbed rerun."""
    import threading
    from ml_conformer_generator_amd.egnn import EGNNDynamics
    d = EGNNDynamics(device=DEV)
    d.load_reference_state_dict(edm_sd)
    d2 = EGNNDynamics(device=DEV)
    d2.load_reference_state_dict(edm_sd)
    stop, errors, churned = threading.Event(), [], [0]

    def churn():
        g = torch.Generator().manual_seed(99)
        try:
            while not stop.is_set():
                sz = torch.randint(6, 30, (int(torch.randint(2, 9, (1,), generator=g)),), generator=g)
                N = int(sz.max())
                nm = (torch.arange(N).unsqueeze(0) < sz.unsqueeze(1)).float().unsqueeze(2).to(DEV)
                plan = d2.plan(sz, N)
                z = torch.randn(len(sz), N, 11, device=DEV) * nm
                d2.run(plan, torch.full((len(sz),), 0.5, device=DEV), z, torch.zeros(len(sz), N, 3, device=DEV))
                d2._plans.clear()                    # drops the only reference: BatchPlan.__del__ -> mcg_plan_destroy on THIS thread
                del plan
                churned[0] += 1
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    th = threading.Thread(target=churn, daemon=True)
    th.start()
    g = torch.Generator().manual_seed(5)
    cases = []
    try:
        for k in range(12):
            sz = torch.randint(9, 28, (6,), generator=g)
            N = int(sz.max())
            nm = (torch.arange(N).unsqueeze(0) < sz.unsqueeze(1)).float().unsqueeze(2)
            z = (torch.randn(6, N, 11, generator=g) * nm).to(DEV)
            ctx = (torch.randn(6, 1, 3, generator=g).repeat(1, N, 1) * nm).to(DEV)
            t = torch.full((6,), 0.1 + 0.07 * k, device=DEV)
            plan = d.plan(sz, N)
            first = d.run(plan, t, z, ctx).clone()               # plan + HIP-graph capture under churn
            again = d.run(plan, t, z, ctx).clone()               # graph replay under churn
            cases.append((sz, N, z, ctx, t, first, again))
    finally:
        stop.set()
        th.join(timeout=60)
    assert not errors, errors
    assert churned[0] >= 3, churned
    d._plans.clear()
    for sz, N, z, ctx, t, first, again in cases:                 # undisturbed reruns on fresh plans: bit-identical
        ref = d.run(d.plan(sz, N), t, z, ctx)
        assert torch.equal(first, ref) and torch.equal(again, ref)
