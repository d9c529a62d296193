"""CPU tests of the host-side mirror of the reference interface (no GPU, no HIP calls)."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN as GOLD
from conftest import REPO, load_golden
from ml_conformer_generator_amd import config as C
from ml_conformer_generator_amd import mol_utils as MU
from ml_conformer_generator_amd import schedule as S
from ml_conformer_generator_amd import weights as W
from ml_conformer_generator_amd.handoff import GeneratedMolecule
from oracle import diffusion_oracle as DO
from oracle import host_oracle as HO


def test_gamma_table_matches_reference_tables():
    g = load_golden("schedule.npz")
    for T in (20, 100, 250, 1000):
        assert torch.equal(S.gamma_table(T, 1e-5), g[f"gamma_T{T}"])


def test_checkpoint_layout_counts():
    edm = W.edm_spec()
    assert len(edm) + 1 == 230                                   # + gamma.gamma   
    assert sum(int(torch.tensor(s).prod()) for _, s, _, _ in edm) + 1001 == 23_897_351
    gcn = W.adj_mat_seer_spec()
    assert len(gcn) == 22
    assert sum(int(torch.tensor(s).prod()) for _, s, _, _ in gcn) == 21_800_531
    sd = W.synth_state_dict(edm[:6], 1)
    assert torch.equal(sd[edm[0][0]], W.synth_state_dict(edm[:6], 1)[edm[0][0]])     # deterministic
    with pytest.raises(RuntimeError):
        W.check_state_dict({}, edm, "edm")


def test_masks_and_edm_input_match_reference():
    g = load_golden("edm_input.npz")
    norms = {k: torch.tensor(v) for k, v in C.CONTEXT_NORMS.items()}
    torch.manual_seed(int(g["seed"]))
    nm, em, ctx = MU.prepare_edm_input(6, g["ref_context"], norms, 15, 19, torch.device("cpu"))
    assert torch.equal(nm, g["node_mask"]) and torch.equal(em, g["edge_mask"]) and torch.equal(ctx, g["context"])
    nm2, em2 = HO.masks_from_sizes(nm.sum(1).reshape(-1).long(), 19)
    assert torch.equal(nm, nm2) and torch.equal(em, em2)


def test_context_shape_kats():
    g = load_golden("context_shape.npz")
    for name in ("ceyyag", "yibfeu", "paba", "frag_yibfeu"):
        xyz = g[f"{name}_xyz"]
        c, rot = MU.get_context_shape(xyz - xyz.mean(0))
        assert torch.allclose(c, g[f"{name}_context"], rtol=1e-6, atol=1e-4)
        assert torch.allclose(rot.abs(), g[f"{name}_rotated"].abs(), atol=1e-3)
    assert torch.allclose(MU.distance_matrix(g["paba_xyz"]), HO.pairwise_distance(g["paba_xyz"]))


def test_parse_molblock():
    text = """X
     RDKit          3D

  3  2  0  0  0  0  0  0  0  0999 V2000
    1.0000    2.0000    3.0000 C   0  0  0  0  0  0  0  0  0  0  0  0
   -1.5000    0.2500    0.0000 Cl  0  0  0  0  0  0  0  0  0  0  0  0
    0.0000    0.0000    1.0000 H   0  0  0  0  0  0  0  0  0  0  0  0
  1  2  1  0
  1  3  1  0
M  END"""
    xyz, zs = MU.parse_molblock_heavy_atoms(text)
    assert zs == [6, 17] and xyz.shape == (2, 3) and float(xyz[1, 0]) == -1.5


def test_prepare_fragment_contract():
    g = load_golden("context_shape.npz")
    frag = (g["frag_yibfeu_xyz"], g["frag_yibfeu_z"].tolist())
    zk, fm = MU.prepare_fragment(3, frag, torch.device("cpu"), max_n_nodes=19, min_n_nodes=15)
    assert zk.shape == (3, 19, 11) and fm.shape == (3, 19, 1) and float(fm.sum()) == 3 * 8
    oh = MU.one_hot_classes(frag[1]).float()
    zo, fo = HO.fragment_latent(frag[0], oh, 3, 19, 15)
    assert torch.equal(zk, zo) and torch.equal(fm, fo)
    with pytest.raises(ValueError, match="fewer atoms than minimum"):
        MU.prepare_fragment(1, frag, torch.device("cpu"), max_n_nodes=19, min_n_nodes=8)
    with pytest.raises(ValueError, match="more atoms than the maximum"):
        MU.prepare_fragment(1, frag, torch.device("cpu"), max_n_nodes=8, min_n_nodes=15)


def test_step_scalars_match_oracle_expressions():
    import torch.nn.functional as F
    T = 20
    gamma = S.gamma_table(T, 1e-5)
    orc = DO.SamplerOracle({}, T)
    for s_int in (19, 7, 0):
        s = torch.full([1, 1], fill_value=s_int) / T
        t = (torch.full([1, 1], fill_value=s_int) + 1.0) / T
        g_s, g_t = orc.g(s), orc.g(t)
        a_ts = torch.exp(0.5 * (F.logsigmoid(-g_t) - F.logsigmoid(-g_s)))
        got = S.step_scalars(gamma, s_int, T)
        assert float(got[0]) == float(a_ts)


def test_handoff_oracle_shapes_and_proxy_rules():
    """The oracle's restatement of the hand-off tensors (mol_utils.py:146-194) and of the bond write-back + validity
    substitute (mol_utils.py:197-223) on hand-made cases; the HIP kernels are compared with it in the GPU tests."""
    from oracle import host_oracle as HO
    torch.manual_seed(0)
    B, N = 3, 19
    n_nodes = torch.tensor([15, 19, 17])
    x = torch.randn(B, N, 3) * 2
    h = torch.nn.functional.one_hot(torch.randint(0, 7, (B, N)), 8).float()
    el, dm, am = HO.adj_mat_seer_input(x, h, n_nodes)
    assert el.shape == (B, 42) and dm.shape == (B, 42, 42) and am.shape == (B, 42, 42)
    assert dm.dtype == torch.float32 and el.dtype == torch.long
    assert int((el[0, 15:] != 0).sum()) == 0 and set(el[1, :19].tolist()) <= set(C.ATOMIC_NUMBERS)
    assert torch.equal(dm, dm.transpose(1, 2)) and torch.equal(am, am.transpose(1, 2))
    assert float(torch.diagonal(dm, dim1=1, dim2=2).min()) == 1.0 and float(am.max()) == 1.0
    assert float(dm[0, 15:, :].abs().sum() - (42 - 15)) == 0.0            # only the +I survives on padding
    # ethane-like: two carbons, one single bond -> valid; a carbon with 5 bonds -> invalid
    z = torch.tensor([[6, 6] + [0] * 40])
    bonds = torch.zeros(1, 42, 42, dtype=torch.int8)
    bonds[0, 1, 0] = 1
    bonds[0, 0, 1] = 3                       # upper triangle is ignored (tril, mol_utils.py:210)
    sym, ok = HO.bond_writeback(bonds, z, torch.tensor([2]))
    assert bool(ok[0]) and int(sym[0, 0, 1]) == 1 and int(sym[0, 1, 0]) == 1
    z5 = torch.tensor([[6] * 6 + [0] * 36])
    b5 = torch.zeros(1, 42, 42, dtype=torch.int8)
    b5[0, 1:6, 0] = 1
    assert not bool(HO.bond_writeback(b5, z5, torch.tensor([6]))[1][0])
    bd = torch.zeros(1, 42, 42, dtype=torch.int8)                             # two disconnected atoms
    assert not bool(HO.bond_writeback(bd, z, torch.tensor([2]))[1][0])
    ar = torch.zeros(1, 42, 42, dtype=torch.int8)                             # aromatic bonds count 1.5: 2 on O is too many
    ar[0, 1, 0] = 4; ar[0, 2, 1] = 4
    assert not bool(HO.bond_writeback(ar, torch.tensor([[6, 8, 6] + [0] * 39]), torch.tensor([3]))[1][0])
    assert bool(HO.bond_writeback(ar, torch.tensor([[6, 6, 6] + [0] * 39]), torch.tensor([3]))[1][0])
    m = GeneratedMolecule([6, 8], torch.zeros(2, 3), torch.zeros(2, 2, dtype=torch.int8))
    assert m.symbols == ["C", "O"] and m.to_xyz_block().startswith("2\n\nC 0.000000000")
    bo = torch.tensor([[0, 2, 0], [2, 0, 1], [0, 1, 0]], dtype=torch.int8)
    m3 = GeneratedMolecule([8, 6, 17], torch.tensor([[0.0, 0, 0], [1.2, 0, 0], [2.1, 1.4, 0]]), bo)
    blk = m3.to_molblock("t")
    xyz, zs = MU.parse_molblock_heavy_atoms(blk)                  # round trip through the V2000 parser
    assert zs == [8, 6, 17] and torch.allclose(xyz, m3.coords, atol=1e-4)
    assert "  3  2  0" in blk and "  2  1  2  0" in blk and "  3  2  1  0" in blk and blk.rstrip().endswith("M  END")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "ml_conformer_generator_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f


def test_inertial_fragment_matching_front_end_matches_reference():
    """ifm_prepare_gen_fragment_context / inverse_coord_transform /
    ifm_prepare_fragments_for_merge / shift_moi_to_com_batch against reference outputs."""
    g = load_golden("ifm_front_end.npz")
    norms = {k: torch.tensor(v) for k, v in C.CONTEXT_NORMS.items()}
    f_nm, f_em, f_ctx, shift, rot = MU.ifm_prepare_gen_fragment_context(
        fixed_fragment_x=g["frag_x"], reference_context=g["ref_context"], context_norms=norms, n_nodes=g["n_nodes"],
        max_n_nodes=25, min_n_nodes=21, device=torch.device("cpu"))
    assert torch.equal(f_nm, g["frag_node_mask"]) and float(f_em.sum()) == float(g["frag_edge_mask_sum"])
    assert torch.allclose(shift, g["shift"], rtol=1e-6, atol=1e-6)
    assert torch.allclose(f_ctx, g["frag_context"], rtol=1e-4, atol=1e-4)
    # eigenvectors are defined up to sign: compare the action of the inverse transform up to that freedom
    assert torch.allclose(rot.abs(), g["rotation"].abs(), atol=1e-4)
    back = MU.inverse_coord_transform(coord=g["xg"], shift=g["shift"], rotation=g["rotation"])
    assert torch.allclose(back, g["xg_back"], rtol=1e-6, atol=1e-6)
    fh = MU.one_hot_classes(g["frag_z"].tolist()).float()
    zk, fm = MU.ifm_prepare_fragments_for_merge(fixed_fragment_x=g["frag_x"], fixed_fragment_h=fh,
                                                gen_fragments_x=g["xg_back"], gen_fragments_h=g["hg"],
                                                device=torch.device("cpu"), max_n_nodes=25)
    assert torch.equal(zk, g["z_known"]) and torch.equal(fm, g["fixed_mask"])
    ms = MU.shift_moi_to_com_batch(torch.eye(3).unsqueeze(0).repeat(3, 1, 1) * 50.0, g["shift"],
                                   torch.tensor([13.0, 15.0, 17.0]))
    assert torch.allclose(ms, g["moi_shift"], rtol=1e-6, atol=1e-5)
    with pytest.raises(ValueError, match="fewer atoms than minimum"):
        MU.ifm_prepare_gen_fragment_context(g["frag_x"], g["ref_context"], norms, g["n_nodes"], 25, 8, torch.device("cpu"))


def test_batched_principal_frames_match_reference_on_cpu_tensors():
    """The batched frame construction (clique growth as index tensors, fp64 per-molecule sums) is plain tensor
    algebra around the HIP overlap kernel; on CPU tensors it must reproduce the reference's frames."""
    from conftest import load_golden
    from ml_conformer_generator_amd.cheminformatics import shape_quadrupole_batch
    g = load_golden("shape_quadrupole.npz")
    names = ["ceyyag", "yibfeu", "paba", "walk12", "walk27"]
    N = 27
    coords, nn = torch.zeros(len(names), N, 3), []
    for i, n in enumerate(names):
        x = g[f"xyz_{n}"]
        coords[i, : x.shape[0]] = x
        nn.append(x.shape[0])
    mom, frames = shape_quadrupole_batch(coords, torch.tensor(nn), device="cpu")
    for i, n in enumerate(names):
        assert float((mom[i] - g[f"moments_{n}"]).abs().max()) < 2e-5, n
        assert float((frames[i, : nn[i]] - g[f"frame_{n}"]).abs().max()) < 2e-5, n
        assert float(frames[i, nn[i]:].abs().max() if nn[i] < N else 0.0) == 0.0


def test_ifm_merge_oracle_matches_reference_golden():
    """oracle.ifm_merge_input (inverse_coord_transform + ifm_prepare_fragments_for_merge, mol_utils.py:460-524)
    against the reference's own outputs."""
    from oracle import host_oracle as HO
    g = np.load(os.path.join(GOLD, "ifm_front_end.npz"))
    fx = torch.tensor(g["frag_x"])
    fh = MU.one_hot_classes(g["frag_z"].tolist()).float()
    zk, fm = HO.ifm_merge_input(fx, fh, torch.tensor(g["xg"]), torch.tensor(g["hg"]), torch.tensor(g["shift"]),
                                torch.tensor(g["rotation"]), int(g["z_known"].shape[1]))
    assert torch.equal(fm, torch.tensor(g["fixed_mask"]))
    assert float((zk - torch.tensor(g["z_known"])).abs().max()) <= 1e-6


def test_div100_three_fma_form_equals_ieee_division(tmp_path):
    """`mcg_div100` (x * fl(0.01) + one residual correction, csrc/mcg_common.h) against x / 100.0f on every 61st fp32
    bit pattern (the full walk is `tools/native/div100_check.c` with stride 1): bit-identical in the working range."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    exe = str(tmp_path / "div100_check")
    src = os.path.join(REPO, "tools", "native", "div100_check.c")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-fopenmp", src, "-o", exe, "-lm"], check=True)
    r = subprocess.run([exe, "61"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert " 0 of them" in r.stdout


def _check_tables(sizes, N, cus, edge_mt=1, four_tile_units=0):
    import ctypes as C
    from ml_conformer_generator_amd import _lib
    L = _lib.lib()
    a = np.asarray(sizes, dtype=np.int32)
    info = np.zeros(8, dtype=np.int32)
    opts = _lib.PlanOpts(edge_mt=edge_mt, four_tile_units=four_tile_units)
    rc = L.mcg_plan_check_tables(len(a), int(N), a.ctypes.data, C.byref(opts), cus, info.ctypes.data)
    return rc, info.tolist(), (L.mcg_last_error().decode() if rc else "")


def _random_batches(seed=99, trials=24):
    g = torch.Generator().manual_seed(seed)
    for trial in range(trials):
        B = int(torch.randint(1, 40, (1,), generator=g))
        lo, hi = [(2, 9), (6, 42), (15, 39), (1, 4), (27, 27), (40, 42)][trial % 6]
        sizes = torch.randint(lo, hi + 1, (B,), generator=g).tolist()
        yield trial, sizes, max(sizes), [1, 2, 4, 16, 256][trial % 5]


@pytest.mark.parametrize("tail", [None, "all", -1, 8, 24])
def test_plan_unit_tables_are_consistent_for_random_batches(tail):
    """The host half of mcg_plan_create (no GPU): for random batch compositions, device sizes (`cus`) and splits between
    four-tile and quarter-tile units, the library re-derives from the row table which (unit, tile, segment) every edge row
    falls into and checks it against the tables the kernels read: the row's slot is listed by its atom, no slot serves
    two atoms, every slot is written, a four-tile unit parks <= 16 rows, an atom lists at most as many slots as the
    table set declares (<= 4)."""
    from ml_conformer_generator_amd import _lib
    four_tile = 0 if tail is None else _lib.ALL_FOUR_TILE if tail == "all" else tail
    for trial, sizes, N, cus in _random_batches():
        rc, info, err = _check_tables(sizes, N, cus, four_tile_units=four_tile)
        assert rc == 0, (trial, sizes, cus, err)
        n_sets, units0, full0, slots0, span0, units1, slots1, span1 = info
        rows = sum(n * (n - 1) for n in sizes)
        if rows == 0:
            continue
        tiles = (rows + 15) // 16
        wg_all = (tiles + 3) // 4
        if n_sets == 0:          # no workgroup-level tables (an atom's rows span more than four units): per-wave partial path
            assert max(sizes) > 42
            continue
        assert 1 <= n_sets <= 2 and 2 <= span0 <= 4
        assert units0 == full0 + (tiles - 4 * full0 if full0 < wg_all else 0)
        if n_sets == 2:
            assert units1 == wg_all and span1 == 2 and slots1 <= slots0
        if tail == "all" and n_sets == 1:      # four-tile units only - unless 64 rows would touch more than 16 (tiny) atoms
            assert full0 == wg_all or (full0 == 0 and min(sizes) <= 6)


def test_plan_builder_under_address_and_ub_sanitizers(tmp_path):
    """SURVEY.md section 5 (host ASan on the shim): the HOST half of mcg_plan_create_ex - mcg_plan_host.cpp, plain C++ -
    built with -fsanitize=address,undefined by `make asan` (CPU only; GPU sanitizers are not available on this pool) and
    driven over the random batches of the test above x every four-tile / quarter-tile split, plus the refused cases.
    Any out-of-range access in the table construction (the round-1 advisor found one) aborts the driver."""
    import shutil
    import subprocess
    csrc = os.path.join(REPO, "ml_conformer_generator_amd", "csrc")
    exe = os.path.join(REPO, "tools", "native", "plan_host_check")
    if shutil.which("make") is None:
        pytest.skip("no make")
    r = subprocess.run(["make", "-C", csrc, "asan"], capture_output=True, text=True)
    if r.returncode != 0:
        if "sanitizer" in r.stderr or "asan" in r.stderr or "cannot find" in r.stderr or "No such file" in r.stderr:
            pytest.skip("this toolchain has no AddressSanitizer runtime: " + r.stderr[-300:])
        raise AssertionError(r.stderr[-2000:])
    lines = []
    for four_tile in (0, 0x3fffffff, -1, 8, 24):
        for trial, sizes, N, cus in _random_batches(seed=1234, trials=30):
            lines.append(" ".join(map(str, [N, cus, 1, four_tile, 1] + sizes)))
    lines.append(" ".join(map(str, [27, 256, 1, 0, 1] + [27] * 64)))                  # configs[1]
    lines.append(" ".join(map(str, [42, 256, 0, 0, 1] + [42] * 6)))
    lines.append(" ".join(map(str, [39, 256, 4, 0, 1] + [15 + (7 * i) % 25 for i in range(256)])))     # 64-row units
    lines.append(" ".join(map(str, [2, 256, 4, 0, 0] + [2] * 33)))                    # refused: > 16 atoms per 64-row unit
    lines.append(" ".join(map(str, [5, 256, 1, 0, 0] + [6, 3])))                      # refused: n_nodes > N
    lines.append(" ".join(map(str, [60, 256, 1, 0, 1] + [60, 55, 47])))               # no unit tables (N > ~50): row table only
    lines.append(" ".join(map(str, [60, 256, 1, 0, 1] + [10] * 7 + [60])))            # advisor r3: the last molecule owns most rows
    f = tmp_path / "batches.txt"
    f.write_text("\n".join(lines) + "\n")
    r = subprocess.run([exe, str(f)], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    assert f"checked {len(lines)} batches" in r.stdout and " 0 unexpected outcomes" in r.stdout


def test_plan_unit_tables_at_the_bench_shapes():
    """configs[1] on a 256-CU device: one complete round of 512 four-tile workgroups + 760 quarter-tile ones, atoms own at
    most three rows of the sums; the alternative set is four-tile only (702 units, two rows per atom)."""
    rc, info, err = _check_tables([27] * 64, 27, 256)
    assert rc == 0, err
    assert info == [2, 1272, 512, info[3], 3, 702, info[6], 2]
    assert info[6] < info[3] < 2 * info[6]
    # 42-atom molecules: 41 rows per atom run through up to four one-tile units
    rc, info, err = _check_tables([42] * 6, 42, 256)
    assert rc == 0, err
    assert info[2] == 0 and info[4] == 4
    # wide units cannot hold 33 two-atom molecules (64 row-owning atoms in one 64-row unit): refused, with a message
    rc, info, err = _check_tables([2] * 33, 2, 256, edge_mt=4)
    assert rc != 0 and "edge_mt" in err


def test_finish_without_rdkit_filters_by_proxy_and_warns_once():
    """Return-type contract where RDKit is absent (this container, the GPU boxes): GeneratedMolecule records that pass the
    labelled proxy; `optimise_geometry=True` cannot be honoured and says so once per process."""
    import warnings
    from ml_conformer_generator_amd import conformer_generator as CG
    from ml_conformer_generator_amd import rdkit_finish
    from ml_conformer_generator_amd.handoff import GeneratedMolecule
    if CG.HAVE_RDKIT:
        pytest.skip("RDKit is installed: the reference's own gate runs instead")
    assert not rdkit_finish.have_rdkit()
    mols = [GeneratedMolecule([6, 8], torch.zeros(2, 3), torch.tensor([[0, 1], [1, 0]], dtype=torch.int8), valid=v)
            for v in (True, False, True)]
    CG._WARNED_NO_MMFF[0] = False
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        kept, frac = CG._finish(mols, True)
        kept2, _ = CG._finish(mols, True)
    assert len(kept) == 2 and abs(frac - 2 / 3) < 1e-9 and len(kept2) == 2
    assert sum("optimise_geometry" in str(x.message) for x in w) == 1          # once per process
    assert CG._finish([], True) == ([], 0.0)
    assert "V2000" in mols[0].to_molblock() and "M  END" in mols[0].to_molblock()


def test_lazy_molecule_records_assemble_in_constant_python_work_per_molecule():
    """Multi-GPU readiness: every rank of a sharded call assembles the records of the WHOLE gathered batch; that used to
    cost 36-97 us per molecule (slicing + two clones) = 70-170 ms at 8 x 256.  Lazy views: 2 048 molecules in <= 25 ms
    (best of 3: the bound is on the work, not on a GC pause), and the views read the right slices."""
    import time
    from ml_conformer_generator_amd.handoff import molecules_from_tensors
    g = torch.Generator().manual_seed(5)
    B = 2048
    x = torch.randn(B, 39, 3, generator=g)
    el = torch.randint(6, 10, (B, 42), generator=g).to(torch.int8)
    bd = torch.randint(0, 5, (B, 42, 42), generator=g).to(torch.int8)
    n = torch.randint(15, 40, (B,), generator=g).to(torch.int32)
    v = (torch.rand(B, generator=g) < 0.5).to(torch.uint8)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        mols = molecules_from_tensors(x, el, bd, n, v)
        best = min(best, time.perf_counter() - t0)
    assert best <= 0.025, f"{best * 1e3:.1f} ms for {B} molecules"
    assert len(mols) == B
    for b in (0, 7, 2047):
        k = int(n[b])
        m = mols[b]
        assert m.GetNumAtoms() == k and m.valid == bool(v[b])
        assert m.atomic_numbers == el[b, :k].tolist() and torch.equal(m.coords, x[b, :k])
        assert torch.equal(m.bond_orders, bd[b, :k, :k]) and len(m.symbols) == k
    mols[3].valid = False                      # records stay independent of each other
    assert mols[4].valid == bool(v[4])
    # value equality survives the lazy views (two assemblies of the same tensors compare equal, a changed flag does not)
    again = molecules_from_tensors(x, el, bd, n, v)
    assert mols[7] == again[7] and mols[3] != again[3] and mols[7] != mols[8]


def test_atom_order_provider_plumbing_with_a_fake_provider():
    """`rdkit_order.batch_order_and_connectivity` + the host-side validation of `handoff.py`, driven by fake providers
    (RDKit is absent here; the real `rdkit_provider` follows the reference's call sequence and is labelled untested)."""
    import numpy as np
    from ml_conformer_generator_amd import handoff, rdkit_order
    torch.manual_seed(2)
    B, N = 4, 19
    n_nodes = torch.tensor([15, 19, 17, 16])
    real = (torch.arange(N).unsqueeze(0) < n_nodes.unsqueeze(1)).float().unsqueeze(2)
    x = torch.randn(B, N, 3) * real
    h = torch.nn.functional.one_hot(torch.randint(0, 8, (B, N)), 8).float() * real
    calls = []

    def fake(z, coords):
        calls.append((z, coords.copy()))
        n = len(z)
        if len(calls) == 3:
            return None
        return list(np.roll(np.arange(n), 1)), np.eye(n, k=1, dtype=np.uint8) + np.eye(n, k=-1, dtype=np.uint8)

    order, conn, built = rdkit_order.batch_order_and_connectivity(fake, x, h, n_nodes)
    assert built == [True, True, False, True] and order[2] is None and order[0][0] == 14
    assert conn[2].shape == (17, 17) and int(conn[2].sum()) == 0             # placeholder for the dropped molecule
    zt = [6, 7, 8, 9, 15, 16, 17, 35]
    assert calls[1][0] == [zt[k] for k in h[1].argmax(1).tolist()] and calls[1][1].dtype == np.float64
    assert np.allclose(calls[0][1], x[0, :15].double().numpy()) and calls[0][1].shape == (15, 3)
    ot = handoff._order_tensor(order, n_nodes.tolist(), 42)
    assert ot.dtype == torch.int32 and ot.shape == (4, 42) and ot[2].tolist() == list(range(42)) and int(ot[0, 0]) == 14
    assert ot[0, 15:].tolist() == list(range(15, 42))
    ct = handoff._conn_tensor(conn, n_nodes.tolist(), 42)
    assert ct.dtype == torch.uint8 and int(ct[0, 0, 1]) == 1 and int(ct[0, 15:].sum()) == 0
    with pytest.raises(ValueError, match="permutation"):
        handoff._order_tensor([[0, 0, 1]], [3], 42)
    with pytest.raises(ValueError, match="rows"):
        handoff._order_tensor([[0, 1, 2]], [3, 3], 42)
    with pytest.raises(ValueError, match="symmetric"):
        handoff._conn_tensor([np.array([[0, 1], [0, 0]])], [2], 42)
    # order only / connectivity only / nothing
    o, c, b = rdkit_order.batch_order_and_connectivity(lambda z, xyz: (list(range(len(z)))[::-1], None), x, h, n_nodes)
    assert c is None and o[1] == list(range(18, -1, -1)) and all(b)
    o, c, b = rdkit_order.batch_order_and_connectivity(lambda z, xyz: (None, None), x, h, n_nodes)
    assert o is None and c is None
    with pytest.raises(ValueError, match="every molecule or for none"):
        k = [0]

        def half(z, xyz):
            k[0] += 1
            return None, (np.zeros((len(z), len(z))) if k[0] % 2 else None)
        rdkit_order.batch_order_and_connectivity(half, x, h, n_nodes)
    # the XYZ text and the order property parse exactly as the reference writes / reads them (mol_utils.py:39-51,119-122)
    assert rdkit_order.xyz_block([6, 17], [[0.0, 1.0, -2.5], [1.0, 0.0, 0.0]]) == \
        "2\n\nC 0.000000000 1.000000000 -2.500000000\nCl 1.000000000 0.000000000 0.000000000\n"
    assert rdkit_order.parse_smiles_output_order("[3,0,1,2,]") == [3, 0, 1, 2]
    assert rdkit_order.default_provider() is None or rdkit_order.have_rdkit()


class _RecordingChem:
    """Stand-in for `rdkit.Chem` that records the call sequence of `rdkit_finish.mol_from_record` (RDKit is absent)."""

    class rdchem:
        class BondType:
            SINGLE, DOUBLE, TRIPLE, AROMATIC = "S", "D", "T", "A"

    def __init__(self):
        self.log = []

    def MolFromXYZBlock(self, text):
        self.log.append(("MolFromXYZBlock", text))
        return {"xyz": text}

    def MolToXYZBlock(self, mol):
        self.log.append(("MolToXYZBlock",))
        return mol["xyz"]

    def EditableMol(self, mol):
        chem = self

        class Ed:
            def AddBond(self, i, j, t):
                chem.log.append(("AddBond", i, j, t))

            def GetMol(self):
                return "mol"
        return Ed()


def test_rdkit_finish_follows_the_reference_call_sequence():
    """`redefine_bonds` (mol_utils.py:197-223) restated in `rdkit_finish.mol_from_record`: XYZ text with "%.9f" ->
    Mol -> MolToXYZBlock -> Mol -> EditableMol.AddBond(i, j, bond_type_dict[t]) over the strict lower triangle in (i, j)
    loop order - NOT a mol block (advisor finding of round 3).  Checked against a recording stand-in for `rdkit.Chem`."""
    from ml_conformer_generator_amd import rdkit_finish
    from ml_conformer_generator_amd.handoff import GeneratedMolecule
    bo = torch.tensor([[0, 2, 0, 4], [2, 0, 1, 0], [0, 1, 0, 3], [4, 0, 3, 0]], dtype=torch.int8)
    rec = GeneratedMolecule([8, 6, 17, 7], torch.tensor([[0.0, 0, 0], [1.2, 0, 0], [2.1, 1.4, 0], [0.123456789, -1, 2]]), bo)
    chem = _RecordingChem()
    assert rdkit_finish.mol_from_record(rec, Chem=chem) == "mol"
    names = [c[0] for c in chem.log]
    assert names == ["MolFromXYZBlock", "MolToXYZBlock", "MolFromXYZBlock", "AddBond", "AddBond", "AddBond", "AddBond"]
    assert chem.log[0][1] == rec.to_xyz_block() and "N 0.123456791 -1.000000000 2.000000000" in chem.log[0][1]
    assert chem.log[3:] == [("AddBond", 1, 0, "D"), ("AddBond", 2, 1, "S"), ("AddBond", 3, 0, "A"), ("AddBond", 3, 2, "T")]
    chem2 = _RecordingChem()
    assert rdkit_finish.mol_without_bonds(rec, Chem=chem2) == {"xyz": rec.to_xyz_block()} and len(chem2.log) == 1


# ------------------------------------------------------------------------------------------------ CPU / NUMA placement (round 6)
def _fake_sysfs(tmp_path, gpu_nodes, cpulists, cpu_nodes=2):
    """A sysfs tree like an 8-GPU two-socket box: KFD topology nodes 0..cpu_nodes-1 are CPUs (simd_count 0), then one per GPU
    with its render minor; /sys/class/drm/renderD<minor>/device/numa_node; /sys/devices/system/node/node<k>/cpulist."""
    root = str(tmp_path)
    k = 0
    for _ in range(cpu_nodes):
        d = os.path.join(root, "sys/class/kfd/kfd/topology/nodes", str(k)); os.makedirs(d)
        open(os.path.join(d, "properties"), "w").write("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor -1\n")
        k += 1
    for g, node in enumerate(gpu_nodes):
        d = os.path.join(root, "sys/class/kfd/kfd/topology/nodes", str(k)); os.makedirs(d)
        open(os.path.join(d, "properties"), "w").write(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {128 + g}\n")
        dd = os.path.join(root, "sys/class/drm", f"renderD{128 + g}", "device"); os.makedirs(dd)
        open(os.path.join(dd, "numa_node"), "w").write(f"{node}\n")
        k += 1
    for n, text in enumerate(cpulists):
        d = os.path.join(root, "sys/devices/system/node", f"node{n}"); os.makedirs(d)
        open(os.path.join(d, "cpulist"), "w").write(text + "\n")
    return root


def test_rank_cpus_follow_the_gpus_numa_node_from_sysfs_alone(tmp_path):
    """Round-5 review, multi-GPU readiness: 8 ranks x (torch threads + host-pool workers) on a two-socket host with no
    placement.  `affinity.rank_cpus` reads the GPU's NUMA node and that node's cores from sysfs (no HIP call) and divides
    them between the ranks whose GPUs share the node."""
    from ml_conformer_generator_amd import affinity as A
    assert A.parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and A.parse_cpulist("") == []
    root = _fake_sysfs(tmp_path, gpu_nodes=[0, 0, 0, 0, 1, 1, 1, 1], cpulists=["0-63,128-191", "64-127,192-255"])
    env = {}
    allowed = list(range(256))
    assert A.gpu_render_minors(root) == list(range(128, 136))
    assert [A.gpu_numa_node(g, root, env) for g in range(8)] == [0, 0, 0, 0, 1, 1, 1, 1]
    sets = [A.rank_cpus(r, 8, None, root, allowed, env) for r in range(8)]
    assert all(len(s) == 32 for s in sets)                                   # 128 cores per node / 4 ranks per node
    assert sorted(c for s in sets for c in s) == allowed                      # a partition: no core shared, none idle
    node0 = set(A.numa_cpus(0, root))
    assert all(set(s) <= node0 for s in sets[:4]) and all(not (set(s) & node0) for s in sets[4:])
    assert sets[0] == list(range(0, 32)) and sets[3] == list(range(160, 192)) and sets[4] == list(range(64, 96))
    # fewer ranks than GPUs: a lone rank on GPU 0 gets its whole node; 2 ranks on GPUs 0, 1 share node 0
    assert A.rank_cpus(0, 1, None, root, allowed, env) == sorted(node0)
    assert [len(A.rank_cpus(r, 2, None, root, allowed, env)) for r in range(2)] == [64, 64]
    # the cgroup allows fewer cores: only those are handed out
    assert A.rank_cpus(5, 8, None, root, list(range(64, 72)), env) == [66, 67]
    # *_VISIBLE_DEVICES re-maps device indices: rank 0 of 1 on physical GPU 6 sits on node 1
    assert A.gpu_numa_node(0, root, {"HIP_VISIBLE_DEVICES": "6"}) == 1
    assert A.visible_gpu_minors(root, {"ROCR_VISIBLE_DEVICES": "4,5,6,7", "HIP_VISIBLE_DEVICES": "1"}) == [133]
    assert set(A.rank_cpus(0, 1, None, root, allowed, {"HIP_VISIBLE_DEVICES": "6"})) == set(allowed) - node0
    # an explicit device for this rank (rank 1 drives GPU 5)
    assert set(A.rank_cpus(1, 2, 5, root, allowed, env)) <= set(allowed) - node0
    # unknown topology (numa_node -1, or no sysfs at all): an even split of the allowed cores by local rank - never an error
    root2 = _fake_sysfs(tmp_path / "b", gpu_nodes=[-1, -1], cpulists=["0-15"], cpu_nodes=1)
    assert A.rank_cpus(1, 2, None, root2, list(range(16)), env) == list(range(8, 16))
    assert A.rank_cpus(0, 4, None, str(tmp_path / "nowhere"), list(range(8)), env) == [0, 1]
    assert A.rank_cpus(3, 4, None, str(tmp_path / "nowhere"), [5], env) == [5]


def test_pin_moves_every_thread_and_children_inherit_it():
    """`sched_setaffinity(0, ...)` moves the calling thread only; `affinity.pin` walks /proc/self/task.  Run in a child process
    (it narrows its own mask and must not touch the test runner's)."""
    import subprocess
    import sys
    import textwrap
    code = textwrap.dedent("""
        import os, sys, threading, subprocess, time
        sys.path.insert(0, %r)
        import importlib.util
        spec = importlib.util.spec_from_file_location("aff", %r)
        A = importlib.util.module_from_spec(spec); spec.loader.exec_module(A)
        allowed = sorted(os.sched_getaffinity(0))
        stop = threading.Event(); seen = {}
        def bg():
            stop.wait(); seen["bg"] = sorted(os.sched_getaffinity(0))
        t = threading.Thread(target=bg); t.start()
        want = allowed[:2] if len(allowed) >= 2 else allowed
        got = A.pin(want)
        stop.set(); t.join()
        child = subprocess.run([sys.executable, "-c", "import os; print(sorted(os.sched_getaffinity(0)))"], capture_output=True, text=True)
        print(got == want, seen["bg"] == want, child.stdout.strip() == str(want), A.pin([]) == [])
    """) % (REPO, os.path.join(REPO, "ml_conformer_generator_amd", "affinity.py"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert out.stdout.split() == ["True", "True", "True", "True"], out.stdout + out.stderr
