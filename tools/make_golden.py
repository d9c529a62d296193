#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

Runs only where /root/reference exists (never on the GPU box).  The reference
modules are imported unmodified through a package shell (RDKit is absent: its
modules are stubbed so that the pure-tensor helpers import; nothing RDKit-bound
is executed).  Weights are the deterministic synthetic tensors of
`ml_conformer_generator_amd.weights` loaded into the reference modules through
their own `load_state_dict` - the checkpoint key layout is therefore exercised.

Only inputs, seeds and reference OUTPUTS are written - no reference source.
Usage:  python tools/make_golden.py  [--out tests/golden]
"""
import argparse
import importlib
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
REF_SRC = "/root/reference/src"

from ml_conformer_generator_amd import weights as W            # noqa: E402
from ml_conformer_generator_amd.config import CONTEXT_NORMS      # noqa: E402
from ml_conformer_generator_amd.mol_utils import parse_molblock_heavy_atoms  # noqa: E402


def import_reference():
    sys.path.insert(0, REF_SRC)
    for n in ["rdkit", "rdkit.Chem", "rdkit.Chem.rdDetermineBonds", "rdkit.Chem.rdmolops", "rdkit.Chem.AllChem",
              "rdkit.Chem.MolStandardize", "rdkit.Chem.MolStandardize.rdMolStandardize",
              "rdkit.Chem.rdFingerprintGenerator", "rdkit.DataStructs", "rdkit.DataStructs.cDataStructs",
              "rdkit.Geometry"]:
        sys.modules[n] = MagicMock()
    pkg = types.ModuleType("mlconfgen")
    pkg.__path__ = [REF_SRC + "/mlconfgen"]
    sys.modules["mlconfgen"] = pkg
    egnn = importlib.import_module("mlconfgen.egnn")
    ed = importlib.import_module("mlconfgen.equivariant_diffusion")
    ams = importlib.import_module("mlconfgen.adj_mat_seer")
    mu = importlib.import_module("mlconfgen.utils.mol_utils")
    return egnn, ed, ams, mu


class NoiseTape:
    """Record every torch.randn draw (in call order) made while active."""

    def __init__(self):
        self.draws = []

    def __enter__(self):
        self._orig = torch.randn

        def rec(*a, **k):
            out = self._orig(*a, **k)
            self.draws.append(out.clone())
            return out
        torch.randn = rec
        return self

    def __exit__(self, *exc):
        torch.randn = self._orig

    def flat(self):
        return np.concatenate([d.reshape(-1).numpy() for d in self.draws]).astype(np.float32)


def build_edm(egnn, ed, T, sd):
    dyn = egnn.EGNNDynamics(in_node_nf=9, context_node_nf=3, hidden_nf=420)
    gm = ed.EquivariantDiffusion(dynamics=dyn, in_node_nf=8, timesteps=1000, noise_precision=1e-5)
    gm.load_state_dict(sd)                                   # strict: checks the key layout
    gm.gamma = ed.PredefinedNoiseSchedule(timesteps=T, precision=1e-5)
    gm.time_steps = torch.flip(torch.arange(0, T), dims=[0])
    gm.T = T
    return gm.eval()


def record_steps(gm):
    trace = []
    orig = gm.sample_p_zs_given_zt

    def wrapped(*a, **k):
        z = orig(*a, **k)
        trace.append(z.clone())
        return z
    gm.sample_p_zs_given_zt = wrapped
    return trace


from ml_conformer_generator_amd.synthetic import synth_gcn_inputs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    ap.add_argument("--only", default="", help="write only the fixtures whose file name contains this substring "
                                               "(everything is still computed: the fixtures share RNG state and models)")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(8)
    egnn, ed, ams, mu = import_reference()
    norms = {k: torch.tensor(v) for k, v in CONTEXT_NORMS.items()}
    dummy_ctx = torch.tensor([53.6424, 108.3042, 151.4399])

    def save(name, **kw):
        if args.only and args.only not in name:
            return
        np.savez_compressed(os.path.join(args.out, name), **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v))
                                                              for k, v in kw.items()})
        print("wrote", name, {k: tuple(np.asarray(v.numpy() if torch.is_tensor(v) else v).shape) for k, v in kw.items()})

    # 1. noise-schedule tables (a4)
    save("schedule.npz", **{f"gamma_T{T}": ed.PredefinedNoiseSchedule(T, 1e-5).gamma.detach() for T in (20, 100, 250, 1000)})

    # 2. input construction (a16) + dense edge list pattern (F6)
    torch.manual_seed(7)
    nm, em, ctx = mu.prepare_edm_input(6, dummy_ctx, norms, 15, 19, torch.device("cpu"))
    edges = egnn.EGNNDynamics.get_adj_matrix(3, 2, torch.device("cpu"))
    save("edm_input.npz", seed=7, n_samples=6, min_n=15, max_n=19, ref_context=dummy_ctx,
         node_mask=nm, edge_mask=em, context=ctx, edges_n3_b2=edges)

    # 3. context KATs (a16) from the demo molecules' heavy atoms
    kat = {}
    for name in ("ceyyag", "yibfeu", "paba", "frag_yibfeu"):
        xyz, zs = parse_molblock_heavy_atoms(open(f"/root/reference/assets/demo_files/{name}.mol").read())
        centred = xyz - torch.mean(xyz, dim=0)
        c, rot = mu.get_context_shape(centred)
        kat[f"{name}_xyz"] = xyz
        kat[f"{name}_z"] = torch.tensor(zs)
        kat[f"{name}_context"] = c
        kat[f"{name}_rotated"] = rot
    save("context_shape.npz", **kat)

    # 4. EGNN dynamics operator seam (a7-a13), synthetic weights seed 1234
    sd = W.synth_edm_state_dict(1234)
    gm = build_edm(egnn, ed, 100, sd)
    for tag, B, N, lo, seed, scale in (("b2n20", 2, 20, 16, 11, 1.0), ("b4n19", 4, 19, 15, 12, 1.0),
                                       ("b3n39", 3, 39, 15, 13, 1.0), ("b3n27_x30", 3, 27, 27, 14, 30.0)):
        torch.manual_seed(seed)
        nm, em, ctx = mu.prepare_edm_input(B, dummy_ctx, norms, lo, N, torch.device("cpu"))
        z = gm.sample_combined_position_feature_noise(B, N, nm) * scale
        t = torch.full((B, 1), 51.0) / 100
        with torch.no_grad():
            out = gm.dynamics(t, z, nm, em, ctx)
        save(f"dynamics_{tag}.npz", t=t, xh=z, node_mask=nm, context=ctx, out=out, weight_seed=1234, weight_recipe=np.array("v2"))

    # 4b. a single EquivariantBlock and its GCL internals (a9, a11-a13) - kernel-level pins
    torch.manual_seed(21)
    B, N = 2, 20
    nm, em, ctx = mu.prepare_edm_input(B, dummy_ctx, norms, 16, N, torch.device("cpu"))
    nmf, emf = nm.view(B * N, 1), em.view(B * N * N, 1)
    h_in = torch.randn(B * N, 420) * nmf
    x_in = torch.randn(B * N, 3) * nmf
    x0 = torch.randn(B * N, 3) * nmf
    edges = egnn.EGNNDynamics.get_adj_matrix(N, B, torch.device("cpu"))
    blk = gm.dynamics.egnn.e_block_3
    with torch.no_grad():
        d0, _ = egnn.coord2diff(x0, edges)
        d1, unit = egnn.coord2diff(x_in, edges)
        ea = torch.cat([d1, d0], dim=1)
        h1, m1 = blk.gcl_0(h_in, edges, ea, nmf, emf)
        # GCL internals through the reference's own sub-calls (egnn.py:38-68): gated + masked messages, aggregate
        msg1, _ = blk.gcl_0.edge_model(h_in[edges[0]], h_in[edges[1]], ea, emf)
        _, cat1 = blk.gcl_0.node_model(h_in, edges, msg1)
        h_out, x_out = blk(h_in, x_in, edges, nmf, emf, d0)
    # m1 / msg1 only for the edges of sample 0's node 0 (small pin on the edge MLP itself); agg for every node
    save("block3_b2n20.npz", node_mask=nm, h_in=h_in, x_in=x_in, x0=x0, h_after_gcl0=h1,
         m_gcl0_node0=m1[:N], msg_gcl0_node0=msg1[:N], agg_gcl0=cat1[:, 420:], h_out=h_out, x_out=x_out, block=3,
         weight_seed=1234, weight_recipe=np.array("v2"))

    # 5. full sampler trajectory, config C1 shape (a1-a6): T=20, B=4, N=19
    gm = build_edm(egnn, ed, 20, sd)
    torch.manual_seed(31)
    nm, em, ctx = mu.prepare_edm_input(4, dummy_ctx, norms, 15, 19, torch.device("cpu"))
    trace = record_steps(gm)
    with NoiseTape() as tape, torch.no_grad():
        x, h = gm(nm, em, ctx, 0)
    save("sampler_T20_b4n19.npz", node_mask=nm, context=ctx, noise=tape.flat(), z_trace=torch.stack(trace),
         x=x, h=h, T=20, resample_steps=0, weight_seed=1234, weight_recipe=np.array("v2"))
    e2e_runs = {"e2e_T20_b4n19.npz": dict(node_mask=nm, context=ctx, noise=tape.flat(), x=x, h=h, T=20, weight_seed=1234,
                                          weight_recipe=np.array("v2"))}

    # 5a. a second composed-path run at the BASELINE configs[1] molecule size (8 x 27 atoms, T = 8): see section 8 below.
    #     (recipe "v2": its final coordinates stay at bond-length scale, so the hand-off perceives bonds - 22..148 per
    #     molecule; the contractive "x 0.3" weights end at max|x| ~ 400 with no bond at all, a GCN input of identity
    #     adjacencies that would pin nothing.)
    #     T = 8: a one-ulp change of the context moves the final x by 1.5e-5 of max|x| (T = 12: 4e-4 - the untrained
    #     sampler amplifies rounding differences with every step), far inside the stated 1e-3 trajectory tolerance.
    gm = build_edm(egnn, ed, 8, sd)
    torch.manual_seed(33)
    nm, em, ctx = mu.prepare_edm_input(8, dummy_ctx, norms, 27, 27, torch.device("cpu"))
    with NoiseTape() as tape, torch.no_grad():
        x, h = gm(nm, em, ctx, 0)
    e2e_runs["e2e_T8_b8n27.npz"] = dict(node_mask=nm, context=ctx, noise=tape.flat(), x=x, h=h, T=8, weight_seed=1234,
                                         weight_recipe=np.array("v2"))

    # 5b. resampling variant (T=8, resample_steps=1) - "v2d" weights: see ml_conformer_generator_amd/weights.py
    sd_d = W.synth_edm_state_dict(1234, recipe="v2d")
    gm = build_edm(egnn, ed, 8, sd_d)
    torch.manual_seed(32)
    nm, em, ctx = mu.prepare_edm_input(2, dummy_ctx, norms, 15, 17, torch.device("cpu"))
    trace = record_steps(gm)
    with NoiseTape() as tape, torch.no_grad():
        x, h = gm(nm, em, ctx, 1)
    save("sampler_T8_rs1.npz", node_mask=nm, context=ctx, noise=tape.flat(), z_trace=torch.stack(trace),
         x=x, h=h, T=8, resample_steps=1, weight_seed=1234, weight_recipe=np.array("v2d"))

    # 6. inpaint + merge_fragments (a14) with a synthetic 8-atom fragment (frag_yibfeu heavy atoms)
    fxyz, fz = parse_molblock_heavy_atoms(open("/root/reference/assets/demo_files/frag_yibfeu.mol").read())
    fxyz = fxyz - fxyz.mean(0)
    cls = {6: 0, 7: 1, 8: 2, 9: 3, 15: 4, 16: 5, 17: 6, 35: 7}
    foh = torch.zeros(len(fz), 8, dtype=torch.long)
    for i, zz in enumerate(fz):
        foh[i, cls[zz]] = 1
    B, N = 3, 19
    gm = build_edm(egnn, ed, 5, sd_d)
    torch.manual_seed(41)
    nm, em, ctx = mu.prepare_edm_input(B, dummy_ctx, norms, 15, N, torch.device("cpu"))
    n_f = fxyz.size(0)
    z_known = torch.zeros(B, N, 11)
    z_known[:, :n_f, :3] = fxyz
    z_known[:, :n_f, 3:] = foh.float()
    fixed = torch.zeros(B, N, 1)
    fixed[:, :n_f] = 1.0
    trace = record_steps(gm)
    with NoiseTape() as tape, torch.no_grad():
        x, h = gm.inpaint(nm, em, ctx, z_known, fixed, 1, 3)
    save("inpaint_T5.npz", node_mask=nm, context=ctx, z_known=z_known, fixed_mask=fixed, noise=tape.flat(),
         z_trace=torch.stack(trace), x=x, h=h, T=5, resample_steps=1, blend_power=3, weight_seed=1234, weight_recipe=np.array("v2d"))

    gm = build_edm(egnn, ed, 10, sd)
    torch.manual_seed(42)
    # merge: z_known covers the full molecule (fixed fragment + "generated" remainder)
    zk = torch.randn(B, N, 11) * nm
    zk[:, :n_f, :3] = fxyz
    zk[:, :n_f, 3:] = foh.float()
    trace = record_steps(gm)
    with NoiseTape() as tape, torch.no_grad():
        x, h = gm.merge_fragments(nm, em, fixed, ctx, zk, diffusion_level=10, resample_steps=1, blend_power=3)
    save("merge_T10_L10.npz", node_mask=nm, context=ctx, z_known=zk, fixed_mask=fixed, noise=tape.flat(),
         z_trace=torch.stack(trace), x=x, h=h, T=10, diffusion_level=10, resample_steps=1, blend_power=3,
         weight_seed=1234, weight_recipe=np.array("v2"))
    e2e_runs["e2e_merge_T10_L10.npz"] = dict(node_mask=nm, context=ctx, z_known=zk, fixed_mask=fixed, noise=tape.flat(), x=x, h=h, T=10,
                                             diffusion_level=10, resample_steps=1, blend_power=3, weight_seed=1234,
                                             weight_recipe=np.array("v2"), route=np.array("merge_fragments"))
    # the level > T failure mode (quirk H5)
    try:
        gm.merge_fragments(nm, em, fixed, ctx, zk, diffusion_level=50)
        err = "none"
    except Exception as e:  # noqa: BLE001
        err = type(e).__name__
    save("merge_level_gt_T.npz", error=np.array(err))

    # 6b. inertial-fragment-matching front end: pure-tensor helpers run for real
    ref_xyz, _ = parse_molblock_heavy_atoms(open("/root/reference/assets/demo_files/yibfeu.mol").read())
    ref_ctx, _ = mu.get_context_shape(ref_xyz - ref_xyz.mean(0))
    fx, fzs = parse_molblock_heavy_atoms(open("/root/reference/assets/demo_files/frag_yibfeu.mol").read())
    fx = fx - ref_xyz.mean(0)                   # fragment in the reference's centred frame
    n_nodes = torch.tensor([[21], [23], [25]])
    f_nm, f_em, f_ctx, shift, rot = mu.ifm_prepare_gen_fragment_context(
        fixed_fragment_x=fx, reference_context=ref_ctx, context_norms=norms, n_nodes=n_nodes,
        max_n_nodes=25, min_n_nodes=21, device=torch.device("cpu"))
    torch.manual_seed(61)
    xg = torch.randn(3, 25 - fx.size(0), 3) * f_nm
    hg = torch.nn.functional.one_hot(torch.randint(0, 7, (3, 25 - fx.size(0))), 8).float() * f_nm
    xg_back = mu.inverse_coord_transform(coord=xg, shift=shift, rotation=rot)
    fh = torch.zeros(fx.size(0), 8)
    for i, zz in enumerate(fzs):
        fh[i, cls[zz]] = 1
    zk2, fm2 = mu.ifm_prepare_fragments_for_merge(fixed_fragment_x=fx, fixed_fragment_h=fh, gen_fragments_x=xg_back,
                                                  gen_fragments_h=hg, device=torch.device("cpu"), max_n_nodes=25)
    moi_shift = mu.shift_moi_to_com_batch(torch.eye(3).unsqueeze(0).repeat(3, 1, 1) * 50.0, shift, torch.tensor([13.0, 15.0, 17.0]))
    save("ifm_front_end.npz", ref_context=ref_ctx, frag_x=fx, frag_z=torch.tensor(fzs), n_nodes=n_nodes,
         frag_node_mask=f_nm, frag_edge_mask_sum=f_em.sum(), frag_context=f_ctx, shift=shift, rotation=rot,
         xg=xg, hg=hg, xg_back=xg_back, z_known=zk2, fixed_mask=fm2, moi_shift=moi_shift)

    # 6c. shape Tanimoto (grid part): the reference module needs only numpy + torch
    ss = importlib.import_module("mlconfgen.cheminformatics.shape_similarity")
    mols = {}
    for name in ("ceyyag", "yibfeu", "paba", "crown_6"):
        xyz, _ = parse_molblock_heavy_atoms(open(f"/root/reference/assets/demo_files/{name}.mol").read())
        mols[name] = xyz - xyz.mean(0)
    pi = torch.pi
    angs = [torch.tensor([0.0, 0.0, 0.0]), torch.tensor([pi, 0, 0]), torch.tensor([0, pi, 0]), torch.tensor([0, 0, pi])]
    tan = {}
    for a, b in (("ceyyag", "yibfeu"), ("ceyyag", "ceyyag"), ("yibfeu", "paba"), ("crown_6", "ceyyag")):
        scores = []
        for k, ang in enumerate(angs):
            cb = mols[b] if k == 0 else ss.rotate_coord(coord=mols[b], angles=ang)
            scores.append(ss.tanimoto_score(mols[a], cb))
        tan[f"{a}__{b}"] = torch.tensor(scores, dtype=torch.float64)
    save("shape_tanimoto.npz", alpha=ss.ALPHA, **{f"xyz_{k}": v for k, v in mols.items()}, **tan)

    # 6d. principal shape frames (get_shape_quadrupole_for_molecule) and the orientation search of
    #     evaluate_samples (cheminformatics/pipeline.py:40-85) on them
    g = torch.Generator().manual_seed(77)
    walk = {}
    for n_at in (12, 27):
        stepv = torch.nn.functional.normalize(torch.randn(n_at, 3, generator=g), dim=1) * 1.45
        xyz = torch.cumsum(stepv, 0)
        walk[f"walk{n_at}"] = xyz - xyz.mean(0)
    frames = {}
    allm = dict(mols, **walk)
    for name, xyz in allm.items():
        mom, pts = ss.get_shape_quadrupole_for_molecule(coordinates=xyz)
        frames[name] = (mom, pts)
    ev = {}
    for a, b in (("ceyyag", "yibfeu"), ("ceyyag", "walk27"), ("yibfeu", "paba"), ("walk12", "ceyyag")):
        ref_pf, cand_pf = frames[a][1], frames[b][1]
        best, which = ss.tanimoto_score(ref_pf, cand_pf), 0
        for k, ang in enumerate(angs[1:]):
            sc = ss.tanimoto_score(ref_pf, ss.rotate_coord(coord=cand_pf, angles=ang))
            if sc > best:
                best, which = sc, k + 1
        ev[f"best__{a}__{b}"] = torch.tensor([best, which], dtype=torch.float64)
    save("shape_quadrupole.npz", **{f"xyz_{k}": v for k, v in allm.items()},
         **{f"moments_{k}": v[0] for k, v in frames.items()}, **{f"frame_{k}": v[1] for k, v in frames.items()}, **ev)

    # 7. AdjMatSeer (a15), synthetic weights seed 4321
    gsd = W.synth_adj_mat_seer_state_dict(4321)
    gcn = ams.AdjMatSeer(dimension=42, n_hidden=2048, embedding_dim=64, num_embeddings=36, num_bond_types=5).eval()
    gcn.load_state_dict(gsd)
    el, dm, am = synth_gcn_inputs(4, [15, 17, 27, 39], seed=51)
    with torch.no_grad():
        logits = gcn(el, dm, am)
    top2 = torch.topk(logits, 2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1])
    save("adj_mat_seer_b4.npz", elements=el, dist_mat=dm, adj_mat=am, logits=logits,
         argmax=torch.argmax(logits, -1), margin=margin, weight_seed=4321)
    print("min top-2 margin over all entries:", float(margin.min()))

    # 8. the COMPOSED path of generate_conformers (conformer_generator.py:330-366): reference sampler output (x, h of
    #    sections 5 / 5a, recorded noise tape) -> samples_to_rdkit_mol + prepare_adj_mat_seer_input (mol_utils.py:18-57,
    #    146-194; RDKit is absent, so their tensor half is oracle/host_oracle.py:adj_mat_seer_input with its two labelled
    #    substitutes - covalent-radius connectivity, generation order) -> the REFERENCE's AdjMatSeer.forward -> the
    #    consumer's argmax (mol_utils.py:210).  Pins sampler -> hand-off -> GCN -> bond argmax end to end.
    from oracle import host_oracle as HO
    for name, r in e2e_runs.items():
        n_nodes = r["node_mask"].sum(1).reshape(-1).to(torch.long)
        el, dm, am = HO.adj_mat_seer_input(r["x"], r["h"], n_nodes)
        with torch.no_grad():
            logits = gcn(el, dm, am)
        top2 = torch.topk(logits, 2, dim=-1).values
        margin = top2[..., 0] - top2[..., 1]
        save(name, **r, n_nodes=n_nodes, elements=el, dist_mat=dm, adj_mat=am, logits=logits,
             argmax=torch.argmax(logits, -1), margin=margin, gcn_weight_seed=4321)
        D = el.shape[1]
        inside = (torch.arange(D).view(1, D, 1) < n_nodes.view(-1, 1, 1)) & (torch.arange(D).view(1, 1, D) < n_nodes.view(-1, 1, 1))
        print(name, "bonds perceived per molecule:", (am.sum((1, 2)) - D).tolist(), " min margin inside molecules:",
              float(margin[inside].min()), " max|x|:", float(r["x"].abs().max()))

    # 9. the hand-off with the two RDKit-owned decisions INJECTED (`canonicalise`, mol_utils.py:110-126, and
    #    the MolGraph adjacency of `prepare_adj_mat_seer_input`, :146-194): RDKit is absent, so the atom order and the
    #    1-order connectivity are fixed seeded inputs here, applied the way the reference applies RDKit's
    #    (`RenumberAtoms(mol, order)`: new atom p = old atom order[p]; connectivity perceived BEFORE the renumbering), the
    #    distance half through the REFERENCE's own `distance_matrix` (:129-143), then the REFERENCE's AdjMatSeer.forward.
    #    AdjMatSeer is not permutation-equivariant (nodes_coord_fc, adj_mat_seer.py:135-138): these fixtures pin that the
    #    product feeds it the atoms in the order it is given.
    from ml_conformer_generator_amd.config import ATOMIC_NUMBERS
    rcov = {6: 0.76, 7: 0.71, 8: 0.66, 9: 0.57, 15: 1.07, 16: 1.05, 17: 1.02, 35: 1.20}

    def tensor_half(x, h, n_nodes, order, conn, D=42):
        B = x.size(0)
        el_b = torch.zeros(B, D, dtype=torch.long)
        dm_b, am_b = torch.zeros(B, D, D), torch.zeros(B, D, D)
        for b in range(B):
            n = int(n_nodes[b])
            cls_b = torch.argmax(h[b], dim=1)
            z_gen = [ATOMIC_NUMBERS[int(cls_b[i])] for i in range(n)]
            coord = torch.tensor([[float("%.9f" % float(x[b, i, k])) for k in range(3)] for i in range(n)],
                                 dtype=torch.float64)
            perm = torch.as_tensor(order[b][:n], dtype=torch.long)
            coord_p = coord[perm]
            dist = mu.distance_matrix(coord_p)                                   # the reference's function
            pad = torch.nn.functional.pad(dist, (0, D - n, 0, D - n), "constant", 0) + torch.eye(D)
            sc = torch.zeros(D, D)
            sc[:n, :n] = torch.as_tensor(conn[b][:n, :n]).float()[perm][:, perm]
            sc = sc + torch.eye(D)
            sc[sc > 0] = 1
            el_b[b, :n] = torch.tensor([z_gen[int(i)] for i in perm])
            dm_b[b] = pad
            am_b[b] = sc
        return el_b, dm_b, am_b

    def cov_rule(x, h, n_nodes, D=42):
        conn = np.zeros((x.size(0), D, D), dtype=np.uint8)
        for b in range(x.size(0)):
            n = int(n_nodes[b])
            cls_b = torch.argmax(h[b], dim=1)
            r = torch.tensor([rcov[ATOMIC_NUMBERS[int(cls_b[i])]] for i in range(n)], dtype=torch.float64)
            coord = torch.tensor([[float("%.9f" % float(x[b, i, k])) for k in range(3)] for i in range(n)],
                                 dtype=torch.float64)
            c = (mu.distance_matrix(coord) < 1.3 * (r.unsqueeze(0) + r.unsqueeze(1))) & ~torch.eye(n, dtype=torch.bool)
            conn[b, :n, :n] = c.numpy()
        return conn

    r = e2e_runs["e2e_T20_b4n19.npz"]
    n_nodes = r["node_mask"].sum(1).reshape(-1).to(torch.long)
    B9, D9 = r["x"].size(0), 42
    ident = np.tile(np.arange(D9, dtype=np.int32), (B9, 1))
    conn_cov = cov_rule(r["x"], r["h"], n_nodes)
    el, dm, am = tensor_half(r["x"], r["h"], n_nodes, ident, conn_cov)
    save("handoff_tensor_half.npz", x=r["x"], h=r["h"], n_nodes=n_nodes, conn_cov=conn_cov, elements=el, dist_mat=dm,
         adj_mat=am)
    g9 = torch.Generator().manual_seed(909)
    order = ident.copy()
    conn_inj = conn_cov.copy()
    for b in range(B9):
        n = int(n_nodes[b])
        order[b, :n] = torch.randperm(n, generator=g9).numpy()
        flips = torch.randint(0, n, (6, 2), generator=g9)                         # an "external" perception: 6 toggled pairs
        for i, j in flips.tolist():
            if i != j:
                conn_inj[b, i, j] ^= 1
                conn_inj[b, j, i] = conn_inj[b, i, j]
    assert any((order[b, :int(n_nodes[b])] != np.arange(int(n_nodes[b]))).any() for b in range(B9))
    el, dm, am = tensor_half(r["x"], r["h"], n_nodes, order, conn_inj)
    with torch.no_grad():
        logits = gcn(el, dm, am)
        el0, dm0, am0 = tensor_half(r["x"], r["h"], n_nodes, ident, conn_inj)
        logits_gen_order = gcn(el0, dm0, am0)
    top2 = torch.topk(logits, 2, dim=-1).values
    margin = top2[..., 0] - top2[..., 1]
    # how far the GCN is from permutation equivariance on this input: un-permute the permuted run's argmax and compare
    am_p, am_g = torch.argmax(logits, -1), torch.argmax(logits_gen_order, -1)
    moved = 0
    for b in range(B9):
        n = int(n_nodes[b])
        inv = np.argsort(order[b, :n])
        back = am_p[b][:n, :n][inv][:, inv]
        moved += int((back != am_g[b][:n, :n]).sum())
    print("e2e_perm: bond entries that differ between canonical-order and generation-order GCN input:", moved)
    coords_perm = torch.zeros_like(r["x"])
    for b in range(B9):
        n = int(n_nodes[b])
        coords_perm[b, :n] = r["x"][b, torch.as_tensor(order[b, :n], dtype=torch.long)]
    save("e2e_perm_T20_b4n19.npz", **r, n_nodes=n_nodes, order=order, conn_in=conn_inj, elements=el, dist_mat=dm, adj_mat=am,
         x_perm=coords_perm, logits=logits, argmax=torch.argmax(logits, -1), margin=margin,
         order_dependent_entries=moved, gcn_weight_seed=4321)

    # 10. the JUDGED step counts (round 5): BASELINE configs[1] runs T = 100 (101 denoiser calls), configs[4] T = 250 with
    #     fixed-fragment inpainting and resample_steps = 1 (501 calls) - every fixture above is T <= 20.  The reference's own
    #     `EquivariantDiffusion.forward` / `.inpaint` at those lengths under a recorded noise tape, with every 10th / 50th
    #     latent of the trajectory.  Weights: the CONTRACTIVE recipe (nn.Linear-family init x 0.3, what bench.py's fragment
    #     modes time).  The untrained "v2d" network amplifies a one-ulp change of the context to 2.5 % of max|x| over 100
    #     steps IN THE REFERENCE ITSELF - no implementation can be pinned to such a trajectory - while the contractive one
    #     moves by ~3e-7 of max|x|; both factors are measured here and stored.  What these fixtures pin is the ancestral loop
    #     at its real length (schedule lookups at every level, the draw order and count of 101 / 751 noise tensors, the
    #     blend / resample arithmetic of 250 levels); the network's internals are pinned by the "v2" fixtures above.
    def amplification(run, x_ref):
        x_pert = run(True)
        return float((x_ref - x_pert).abs().max() / x_ref.abs().max())

    sd_c = W.synth_edm_state_dict(1234, weight_gain=0.3)

    def run_T100(sd_run, perturb, keep=None):
        gm_ = build_edm(egnn, ed, 100, sd_run)
        torch.manual_seed(51)
        nm_, em_, ctx_ = mu.prepare_edm_input(2, dummy_ctx, norms, 27, 27, torch.device("cpu"))
        if perturb:
            ctx_ = torch.nextafter(ctx_, ctx_ + 1) * nm_                 # one ulp up, every context entry
        tr = record_steps(gm_) if keep is not None else None
        with NoiseTape() as tp, torch.no_grad():
            x_, h_ = gm_(nm_, em_, ctx_, 0)
        if keep is not None:
            keep.update(nm=nm_, ctx=ctx_, tape=tp.flat(), trace=torch.stack(tr), h=h_)
        return x_

    k100 = {}
    x = run_T100(sd_c, False, k100)
    amp_c = amplification(lambda p: run_T100(sd_c, p), x)
    x_v2d = run_T100(sd_d, False)
    amp_d = amplification(lambda p: run_T100(sd_d, p), x_v2d)
    idx = list(range(9, 100, 10))                                        # every 10th z_s (0-based call index)
    print(f"e2e_T100: one-ulp context perturbation moves the final x by {amp_c:.3e} of max|x| (contractive), {amp_d:.3e} (v2d)")
    save("e2e_T100_b2n27.npz", node_mask=k100["nm"], context=k100["ctx"], noise=k100["tape"], z_trace=k100["trace"][idx],
         z_trace_index=np.array(idx), x=x, h=k100["h"], T=100, resample_steps=0, weight_seed=1234,
         weight_recipe=np.array("gain0.3"), one_ulp_context_rel_dev=amp_c, one_ulp_context_rel_dev_v2d=amp_d)

    def run_T250(perturb, keep=None):
        gm_ = build_edm(egnn, ed, 250, sd_c)
        torch.manual_seed(52)
        nm_, em_, ctx_ = mu.prepare_edm_input(2, dummy_ctx, norms, 15, 19, torch.device("cpu"))
        if perturb:
            ctx_ = torch.nextafter(ctx_, ctx_ + 1) * nm_
        zk = torch.zeros(2, 19, 11)
        zk[:, :n_f, :3] = fxyz
        zk[:, :n_f, 3:] = foh.float()
        fm = torch.zeros(2, 19, 1)
        fm[:, :n_f] = 1.0
        tr = record_steps(gm_) if keep is not None else None
        with NoiseTape() as tp, torch.no_grad():
            x_, h_ = gm_.inpaint(nm_, em_, ctx_, zk, fm, 1, 3)
        if keep is not None:
            keep.update(nm=nm_, ctx=ctx_, tape=tp.flat(), trace=torch.stack(tr), h=h_, zk=zk, fm=fm)
        return x_

    k250 = {}
    x = run_T250(False, k250)
    amp_i = amplification(run_T250, x)
    n_tr = int(k250["trace"].shape[0])
    idx = list(range(49, n_tr, 50))
    print(f"inpaint_T250: {n_tr} recorded sampler steps, one-ulp context perturbation moves the final x by {amp_i:.3e} of max|x|")
    save("inpaint_T250_rs1_b2.npz", node_mask=k250["nm"], context=k250["ctx"], z_known=k250["zk"], fixed_mask=k250["fm"],
         noise=k250["tape"], z_trace=k250["trace"][idx], z_trace_index=np.array(idx), n_sampler_steps=n_tr, x=x, h=k250["h"],
         T=250, resample_steps=1, blend_power=3, weight_seed=1234, weight_recipe=np.array("gain0.3"),
         one_ulp_context_rel_dev=amp_i)


if __name__ == "__main__":
    main()
