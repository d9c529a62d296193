"""Runs ONLY where RDKit is installed (it is not in the build container nor on the GPU boxes of this project): the two
RDKit-bound modules against real RDKit.  Everything here is skipped offline; the plumbing around these modules is covered
by tests/test_host_logic.py with stand-ins."""
import numpy as np
import pytest
import torch

Chem = pytest.importorskip("rdkit.Chem")


def _ethanol():
    z = [6, 6, 8]
    xyz = np.array([[-0.887, 0.175, -0.013], [0.460, -0.515, -0.046], [1.443, 0.428, 0.270]], dtype=np.float64)
    return z, xyz


def test_rdkit_provider_returns_a_permutation_and_a_symmetric_connectivity():
    from ml_conformer_generator_amd import rdkit_order
    z, xyz = _ethanol()
    order, conn = rdkit_order.rdkit_provider(z, xyz)
    assert sorted(order) == [0, 1, 2]
    assert conn.shape == (3, 3) and (conn == conn.T).all() and conn[0, 1] == 1 and conn[1, 2] == 1 and conn[0, 2] == 0
    # the order is what the reference's canonicalise() applies: RenumberAtoms(mol, order) puts old atom order[p] at position p
    mol = Chem.MolFromXYZBlock(rdkit_order.xyz_block(z, xyz))
    from rdkit.Chem import rdDetermineBonds
    rdDetermineBonds.DetermineConnectivity(mol)
    Chem.MolToSmiles(mol)
    ren = Chem.RenumberAtoms(mol, order)
    assert [a.GetAtomicNum() for a in ren.GetAtoms()] == [z[i] for i in order]
    with pytest.raises(ValueError):
        rdkit_order.rdkit_provider([6, 6], np.array([[0.0, 0, 0], [9.0, 0, 0]]))          # no bond perceived: the reference raises


def test_rdkit_finish_builds_the_mol_like_redefine_bonds():
    from ml_conformer_generator_amd import rdkit_finish
    from ml_conformer_generator_amd.handoff import GeneratedMolecule
    z, xyz = _ethanol()
    bo = torch.tensor([[0, 1, 0], [1, 0, 1], [0, 1, 0]], dtype=torch.int8)
    rec = GeneratedMolecule(z, torch.tensor(xyz, dtype=torch.float32), bo)
    mol = rdkit_finish.mol_from_record(rec)
    assert mol.GetNumAtoms() == 3 and mol.GetNumBonds() == 2
    assert not any(a.GetIsAromatic() for a in mol.GetAtoms())
    out = rdkit_finish.finish([rec], optimise_geometry=False)
    assert out[0] is not None and Chem.MolToSmiles(out[0]) == "CCO"
    assert rdkit_finish.samples([rec])[0].GetNumBonds() == 0


@pytest.mark.gpu
def test_evaluate_samples_wrapper_runs_where_rdkit_and_a_gpu_exist():
    from ml_conformer_generator_amd.cheminformatics import evaluate_samples
    from rdkit.Chem import AllChem
    ref = Chem.AddHs(Chem.MolFromSmiles("CCOC(=O)c1ccccc1"))
    AllChem.EmbedMolecule(ref, randomSeed=1)
    other = Chem.AddHs(Chem.MolFromSmiles("CCOC(=O)c1ccccn1"))
    AllChem.EmbedMolecule(other, randomSeed=2)
    block, res = evaluate_samples(ref, [ref, other])
    assert "V2000" in block and len(res) == 2
    assert abs(res[0]["chemical_tanimoto"] - 1.0) < 1e-9 and res[0]["shape_tanimoto"] > 0.99
    assert 0.0 < res[1]["chemical_tanimoto"] < 1.0 and 0.0 < res[1]["shape_tanimoto"] <= 1.0


def test_pooled_rdkit_stages_equal_the_serial_ones():
    """f2 (round 5): the two RDKit stages through `host_pool` worker processes (fresh interpreters holding numpy + RDKit
    only) against the in-process path: same order / connectivity, same standardised molecules, same order, `None` for a
    rejected molecule, ValueError for a molecule without a perceived bond."""
    from ml_conformer_generator_amd import host_pool as HP
    from ml_conformer_generator_amd import rdkit_finish, rdkit_order
    from ml_conformer_generator_amd.handoff import GeneratedMolecule
    z, xyz = _ethanol()
    B, N = 12, 3
    x = torch.tensor(xyz, dtype=torch.float32).unsqueeze(0).repeat(B, 1, 1) + 0.01 * torch.arange(B).view(B, 1, 1)
    h = torch.nn.functional.one_hot(torch.tensor([0, 0, 2]), 8).float().unsqueeze(0).repeat(B, 1, 1)
    n = torch.full((B,), N)
    with HP.HostPool(3) as pool:
        serial = rdkit_order.batch_order_and_connectivity(rdkit_order.rdkit_provider, x, h, n)
        pooled = rdkit_order.batch_order_and_connectivity(rdkit_order.rdkit_provider, x, h, n, pool)
        assert serial[0] == pooled[0] and serial[2] == pooled[2]
        assert all(np.array_equal(a, b) for a, b in zip(serial[1], pooled[1]))
        bo = torch.tensor([[0, 1, 0], [1, 0, 1], [0, 1, 0]], dtype=torch.int8)
        recs = [GeneratedMolecule(z, x[b], bo) for b in range(B)]
        recs[5] = GeneratedMolecule(z, x[5], torch.tensor([[0, 3, 3], [3, 0, 3], [3, 3, 0]], dtype=torch.int8))   # valence: rejected
        for opt in (False, True):
            a = rdkit_finish.finish(recs, optimise_geometry=opt)
            b = rdkit_finish.finish(recs, optimise_geometry=opt, executor=pool)
            assert [m is None for m in a] == [m is None for m in b] and a[5] is None
            for ma, mb in zip(a, b):
                if ma is not None:
                    assert Chem.MolToSmiles(ma) == Chem.MolToSmiles(mb)
                    assert np.allclose(ma.GetConformer().GetPositions(), mb.GetConformer().GetPositions(), atol=1e-6)
        assert len(rdkit_finish.samples(recs, pool)) == B
        x_far = x.clone(); x_far[7] = x_far[7] * 40.0
        with pytest.raises(ValueError):
            rdkit_order.batch_order_and_connectivity(rdkit_order.rdkit_provider, x_far, h, n, pool)
