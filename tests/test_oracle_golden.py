"""Pin the CPU oracle against outputs of the reference itself (tests/golden, made by
tools/make_golden.py).  CPU-only."""
import os

import numpy as np
import pytest
import torch

from conftest import TapeNoise, load_golden, sd_for
from oracle import diffusion_oracle as DO
from oracle import egnn_oracle as EO
from oracle import gcn_oracle as GO
from oracle import host_oracle as HO
from parity_tolerance import LONG_TRAJ_REL, traj_violation, violation

torch.set_num_threads(8)


def edge_mask_of(node_mask):
    B, N, _ = node_mask.shape
    nm = node_mask.squeeze(2)
    em = nm.unsqueeze(1) * nm.unsqueeze(2) * (1 - torch.eye(N)).unsqueeze(0)
    return em.reshape(B * N * N, 1)


def test_gamma_tables_bit_exact():
    g = load_golden("schedule.npz")
    for T in (20, 100, 250, 1000):
        assert torch.equal(DO.gamma_schedule(T, 1e-5), g[f"gamma_T{T}"])
    # survey KATs
    t20 = DO.gamma_schedule(20, 1e-5)
    assert abs(float(t20[0]) - (-11.5115576)) < 1e-5 and abs(float(t20[20]) - 10.8447676) < 1e-5


def test_input_construction():
    g = load_golden("edm_input.npz")
    torch.manual_seed(int(g["seed"]))
    norms = {"mean": torch.tensor([105.0766, 473.1938, 537.4675]), "mad": torch.tensor([52.0409, 219.7475, 232.9718])}
    nm, em, ctx = HO.edm_input(6, g["ref_context"], norms, 15, 19)
    assert torch.equal(nm, g["node_mask"]) and torch.equal(em, g["edge_mask"]) and torch.equal(ctx, g["context"])
    r, c = EO.dense_edge_index(3, 2)
    assert torch.equal(torch.stack([r, c]), g["edges_n3_b2"])


def test_context_kats():
    g = load_golden("context_shape.npz")
    for name in ("ceyyag", "yibfeu", "paba", "frag_yibfeu"):
        xyz = g[f"{name}_xyz"]
        c, rot = HO.context_shape(xyz - xyz.mean(0))
        assert torch.allclose(c, g[f"{name}_context"], rtol=1e-6, atol=1e-4)
    assert torch.allclose(g["ceyyag_context"], torch.tensor([50.589699, 105.313202, 133.522293]), atol=2e-3)


@pytest.mark.parametrize("tag", ["b2n20", "b4n19", "b3n39", "b3n27_x30"])
def test_dynamics_seam(edm_sd, tag):
    g = load_golden(f"dynamics_{tag}.npz")
    out = EO.egnn_dynamics(edm_sd, g["t"], g["xh"], g["node_mask"], edge_mask_of(g["node_mask"]), g["context"])
    assert torch.allclose(out, g["out"], rtol=1e-5, atol=1e-6), float((out - g["out"]).abs().max())


def test_single_block(edm_sd):
    g = load_golden("block3_b2n20.npz")
    nm = g["node_mask"]
    B, N, _ = nm.shape
    nmf, emf = nm.reshape(B * N, 1), edge_mask_of(nm)
    row, col = EO.dense_edge_index(N, B)
    d0, _ = EO.pair_geometry(g["x0"], row, col)
    d1, _ = EO.pair_geometry(g["x_in"], row, col)
    p = "dynamics.egnn.e_block_3."
    h1, m1, msg1, agg1 = EO.gcl(edm_sd, p + "gcl_0.", g["h_in"], row, col, torch.cat([d1, d0], 1), nmf, emf)
    assert torch.allclose(h1, g["h_after_gcl0"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(m1[:N], g["m_gcl0_node0"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(msg1[:N], g["msg_gcl0_node0"], rtol=1e-5, atol=1e-6)      # gate * mask (egnn.py:48-51)
    assert torch.allclose(agg1, g["agg_gcl0"], rtol=1e-5, atol=1e-6)                # segment sum / 100 (egnn.py:59-64)
    assert float(g["agg_gcl0"].abs().max()) > 0.05                                  # the aggregate carries real signal
    h, x = EO.equivariant_block(edm_sd, p, g["h_in"], g["x_in"], row, col, nmf, emf, d0)
    assert torch.allclose(h, g["h_out"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(x, g["x_out"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["sampler_T20_b4n19.npz", "sampler_T8_rs1.npz"])
def test_sampler_trajectory(name):
    g = load_golden(name)
    assert bool(torch.isfinite(g["z_trace"]).all())
    nm = g["node_mask"]
    s = DO.SamplerOracle(sd_for(g), int(g["T"]), noise_fn=TapeNoise(g["noise"]))
    s.trace = []
    x, h = s.forward(nm, edge_mask_of(nm), g["context"], int(g["resample_steps"]))
    assert s.noise_fn.pos == g["noise"].numel()          # same number of draws, same order
    zt = torch.stack(s.trace)
    assert traj_violation(zt, g["z_trace"], rel=2e-5) <= 1.0, traj_violation(zt, g["z_trace"], rel=2e-5)
    assert violation(x, g["x"]) <= 1.0
    assert torch.equal(h.to(torch.int64), g["h"].to(torch.int64))


def test_inpaint():
    g = load_golden("inpaint_T5.npz")
    assert bool(torch.isfinite(g["z_trace"]).all())
    nm = g["node_mask"]
    s = DO.SamplerOracle(sd_for(g), int(g["T"]), noise_fn=TapeNoise(g["noise"]))
    s.trace = []
    x, h = s.inpaint(nm, edge_mask_of(nm), g["context"], g["z_known"], g["fixed_mask"], 1, 3)
    assert s.noise_fn.pos == g["noise"].numel()
    assert traj_violation(torch.stack(s.trace), g["z_trace"], rel=2e-5) <= 1.0
    assert violation(x, g["x"]) <= 1.0
    assert torch.equal(h.to(torch.int64), g["h"].to(torch.int64))


def test_sampler_at_the_judged_step_counts():
    """Round 5: the reference's own `forward` at T = 100 (BASELINE configs[1]: 101 denoiser calls) and `inpaint` at T = 250
    with resample_steps = 1 (configs[4]: 501 calls) under recorded noise tapes, every 10th / 50th latent kept
    (`e2e_T100_b2n27.npz`, `inpaint_T250_rs1_b2.npz`; contractive weights - the fixtures record that a one-ulp change of the
    context moves the reference's own final x by 3e-7 / 6e-7 of max|x| there, and by 2.5 % with the untrained "v2d"
    recipe, which therefore cannot pin anything at this length).  The oracle sampler replays both."""
    g = load_golden("e2e_T100_b2n27.npz")
    assert float(g["one_ulp_context_rel_dev"]) < 1e-5 < 1e-3 < float(g["one_ulp_context_rel_dev_v2d"])
    nm = g["node_mask"]
    s = DO.SamplerOracle(sd_for(g), int(g["T"]), noise_fn=TapeNoise(g["noise"]))
    s.trace = []
    x, h = s.forward(nm, edge_mask_of(nm), g["context"], 0)
    assert s.noise_fn.pos == g["noise"].numel() and len(s.trace) == 100
    zt = torch.stack(s.trace)[g["z_trace_index"].long()]
    assert traj_violation(zt, g["z_trace"], rel=LONG_TRAJ_REL) <= 1.0, traj_violation(zt, g["z_trace"], rel=LONG_TRAJ_REL)
    assert violation(x, g["x"]) <= 1.0
    assert torch.equal(h.to(torch.int64), g["h"].to(torch.int64))
    g = load_golden("inpaint_T250_rs1_b2.npz")
    assert float(g["one_ulp_context_rel_dev"]) < 1e-5
    nm = g["node_mask"]
    # 501 denoiser calls are ~35 s of an idle 8-core host and many minutes of a contended one: by default the replay stops
    # after the 3rd kept latent (151 sampler steps = 152 calls of the resampling loop); MCG_ORACLE_FULL=1 runs all 500 steps
    # and checks the decoded x / h as well.  (The HIP path is compared with this fixture at full length either way:
    # tests/test_hip_parity.py::test_sampler_at_the_judged_step_counts_vs_reference_golden.)
    full = os.environ.get("MCG_ORACLE_FULL", "0") == "1"
    idx = g["z_trace_index"].long()
    keep = idx.numel() if full else 3
    stop_after = None if full else int(idx[keep - 1]) + 1

    class _Enough(Exception):
        pass

    s = DO.SamplerOracle(sd_for(g), int(g["T"]), noise_fn=TapeNoise(g["noise"]))

    class _Trace(list):
        def append(self, z):
            super().append(z)
            if stop_after is not None and len(self) >= stop_after:
                raise _Enough

    s.trace = _Trace()
    try:
        x, h = s.inpaint(nm, edge_mask_of(nm), g["context"], g["z_known"], g["fixed_mask"], 1, 3)
    except _Enough:
        x = h = None
    assert (x is not None) == full
    zt = torch.stack(list(s.trace))[idx[:keep]]
    assert traj_violation(zt, g["z_trace"][:keep], rel=LONG_TRAJ_REL) <= 1.0, traj_violation(zt, g["z_trace"][:keep], rel=LONG_TRAJ_REL)
    if full:
        assert s.noise_fn.pos == g["noise"].numel() and len(s.trace) == int(g["n_sampler_steps"]) == 500
        assert violation(x, g["x"]) <= 1.0
        assert torch.equal(h.to(torch.int64), g["h"].to(torch.int64))


def test_merge_fragments():
    g = load_golden("merge_T10_L10.npz")
    nm = g["node_mask"]
    s = DO.SamplerOracle(sd_for(g), int(g["T"]), noise_fn=TapeNoise(g["noise"]))
    s.trace = []
    x, h = s.merge_fragments(nm, edge_mask_of(nm), g["fixed_mask"], g["context"], g["z_known"], int(g["diffusion_level"]), 1, 3)
    assert s.noise_fn.pos == g["noise"].numel()
    assert traj_violation(torch.stack(s.trace), g["z_trace"], rel=2e-5) <= 1.0
    assert violation(x, g["x"]) <= 1.0
    assert torch.equal(h.to(torch.int64), g["h"].to(torch.int64))
    # diffusion_level > T fails exactly like the reference (quirk H5)
    assert str(load_golden("merge_level_gt_T.npz")["error"]) == "IndexError"
    with pytest.raises(IndexError):
        s.merge_fragments(nm, edge_mask_of(nm), g["fixed_mask"], g["context"], g["z_known"], 50)


def test_adj_mat_seer(gcn_sd):
    g = load_golden("adj_mat_seer_b4.npz")
    logits = GO.adj_mat_seer(gcn_sd, g["elements"], g["dist_mat"], g["adj_mat"])
    assert torch.allclose(logits, g["logits"], rtol=1e-5, atol=1e-5), float((logits - g["logits"]).abs().max())
    assert torch.equal(torch.argmax(logits, -1), g["argmax"])
    assert torch.equal(logits, logits.transpose(1, 2))


@pytest.mark.parametrize("name", ["e2e_T20_b4n19.npz", "e2e_T8_b8n27.npz", "e2e_merge_T10_L10.npz"])
def test_composed_path_oracle_vs_reference(name, gcn_sd):
    """The COMPOSED path of generate_conformers (conformer_generator.py:330-366): sampler (recorded noise tape) ->
    hand-off tensors -> AdjMatSeer -> bond argmax.  The fixture holds what the reference's own EquivariantDiffusion and
    AdjMatSeer produce (tools/make_golden.py section 8); the oracle pipeline must land on the same adjacency."""
    g = load_golden(name)
    nm = g["node_mask"]
    orc = DO.SamplerOracle(sd_for(g), int(g["T"]), noise_fn=TapeNoise(g["noise"]))
    if "route" in g:            # the fragment-merge route of edm_samples (conformer_generator.py:231-240)
        x, h = orc.merge_fragments(nm, edge_mask_of(nm), g["fixed_mask"], g["context"], g["z_known"], int(g["diffusion_level"]),
                                   int(g["resample_steps"]), int(g["blend_power"]))
    else:
        x, h = orc.forward(nm, edge_mask_of(nm), g["context"], 0)
    assert traj_violation(x.unsqueeze(0), g["x"].unsqueeze(0), split=None) <= 1.0
    assert torch.equal(h.to(torch.int64), g["h"].to(torch.int64))
    el, dm, am = HO.adj_mat_seer_input(x, h, g["n_nodes"])
    assert torch.equal(el, g["elements"]) and torch.equal(am, g["adj_mat"])
    assert torch.allclose(dm, g["dist_mat"], rtol=1e-5, atol=1e-4)
    logits = GO.adj_mat_seer(gcn_sd, el, dm, am)
    err = float((logits - g["logits"]).abs().max())
    assert err <= 1e-4 * float(g["logits"].abs().max())
    safe = g["margin"] > 100 * max(err, 1e-9)
    assert torch.equal(logits.argmax(-1)[safe], g["argmax"][safe]) and float(safe.float().mean()) > 0.99
    assert int((am.sum((1, 2)) - 42).min()) > 0                     # every molecule has perceived bonds: a non-trivial GCN input


def test_handoff_distance_half_vs_reference_distance_matrix():
    """`host_oracle.adj_mat_seer_input` against the tensor half built with the REFERENCE's own `distance_matrix`
    (mol_utils.py:129-143; fixture `handoff_tensor_half.npz`, generation order, covalent-radius rule): bit-exact."""
    g = load_golden("handoff_tensor_half.npz")
    el, dm, am = HO.adj_mat_seer_input(g["x"], g["h"], g["n_nodes"])
    assert torch.equal(el, g["elements"]) and torch.equal(dm, g["dist_mat"]) and torch.equal(am, g["adj_mat"])
    el, dm, am = HO.adj_mat_seer_input(g["x"], g["h"], g["n_nodes"], conn=list(g["conn_cov"]))
    assert torch.equal(am, g["adj_mat"])


def test_handoff_with_injected_order_and_connectivity_vs_reference(gcn_sd):
    """The two RDKit-owned decisions injected (`canonicalise`, mol_utils.py:110-126): a fixed non-identity atom order and
    an external connectivity, applied by the fixture generator the way the reference applies RDKit's, distances through
    the reference's `distance_matrix`, logits from the REFERENCE's AdjMatSeer.  The oracle's hand-off is bit-exact on the
    inputs; the oracle GCN reproduces the reference's bonds on the permuted input; and the fixture itself records that the
    GCN is order dependent (hundreds of bond entries change with the atom order)."""
    g = load_golden("e2e_perm_T20_b4n19.npz")
    assert int(g["order_dependent_entries"]) > 100
    order, conn = [r.tolist() for r in g["order"]], list(g["conn_in"])
    el, dm, am, coords = HO.adj_mat_seer_input(g["x"], g["h"], g["n_nodes"], order=order, conn=conn, return_coords=True)
    assert torch.equal(el, g["elements"]) and torch.equal(dm, g["dist_mat"]) and torch.equal(am, g["adj_mat"])
    for b, c in enumerate(coords):
        assert torch.allclose(c.float(), g["x_perm"][b, : c.shape[0]], rtol=0, atol=1e-6)
    logits = GO.adj_mat_seer(gcn_sd, el, dm, am)
    err = float((logits - g["logits"]).abs().max())
    assert err <= 1e-4 * float(g["logits"].abs().max())
    safe = g["margin"] > 100 * max(err, 1e-9)
    assert torch.equal(logits.argmax(-1)[safe], g["argmax"][safe]) and float(safe.float().mean()) > 0.99
    # generation order on the same connectivity gives DIFFERENT elements rows (the permutation is not the identity)
    el0, _, _ = HO.adj_mat_seer_input(g["x"], g["h"], g["n_nodes"], conn=conn)
    assert not torch.equal(el0, el)


def test_shape_tanimoto_oracle_matches_reference():
    import numpy as np
    from oracle import shape_oracle as SO
    g = load_golden("shape_tanimoto.npz")
    assert abs(SO.ALPHA - float(g["alpha"])) < 1e-15
    pi = torch.pi
    angs = (torch.tensor([pi, 0, 0]), torch.tensor([0, pi, 0]), torch.tensor([0, 0, pi]))
    for key in ("ceyyag__yibfeu", "ceyyag__ceyyag", "yibfeu__paba", "crown_6__ceyyag"):
        a, b = key.split("__")
        xa, xb = g["xyz_" + a], g["xyz_" + b]
        mine = [SO.tanimoto_score(xa, xb)] + [SO.tanimoto_score(xa, SO.rotate_coord(xb, ang)) for ang in angs]
        assert np.abs(np.array(mine) - g[key].numpy()).max() < 1e-7, key
    assert abs(SO.tanimoto_score(g["xyz_ceyyag"], g["xyz_ceyyag"]) - 1.0) < 1e-6


def test_shape_quadrupole_oracle_matches_reference():
    """Principal shape frames (get_shape_quadrupole_for_molecule, shape_similarity.py:18-202): the oracle
    restatement against the reference's outputs - same operations in the same order, so the tolerance is tight
    (moments 1e-5; coordinates 1e-5 except crown_6, whose two largest moments differ by < 1 %)."""
    from oracle import shape_oracle as SO
    g = load_golden("shape_quadrupole.npz")
    for name in ("paba", "ceyyag", "crown_6", "walk12", "yibfeu"):
        mom, frame = SO.shape_quadrupole(g[f"xyz_{name}"])
        assert float((mom - g[f"moments_{name}"]).abs().max()) < 1e-5, name
        assert float((frame - g[f"frame_{name}"]).abs().max()) < (1e-3 if name == "crown_6" else 1e-5), name
    # the orientation search of evaluate_samples on principal frames
    for key in ("best__ceyyag__yibfeu", "best__yibfeu__paba"):
        _, a, b = key.split("__")
        best, which = SO.best_orientation_score(g[f"frame_{a}"], g[f"frame_{b}"])
        assert abs(best - float(g[key][0])) < 1e-6 and which == int(g[key][1])
