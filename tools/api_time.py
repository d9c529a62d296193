import sys, time, torch
sys.path.insert(0, '/root/repo')
from ml_conformer_generator_amd import MLConformerGenerator
from ml_conformer_generator_amd import weights as W
from ml_conformer_generator_amd.synthetic import DUMMY_CONTEXT
dev = torch.device('cuda:0')
gen = MLConformerGenerator(diffusion_steps=100, device=dev, edm_weights=W.synth_edm_state_dict(1234), adj_mat_seer_weights=W.synth_adj_mat_seer_state_dict(4321))
ctx = torch.tensor(DUMMY_CONTEXT)
for n in (64, 64, 256):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mols = gen.generate_conformers(reference_context=ctx, n_atoms=27, n_samples=n, variance=0 if n == 64 else 12)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"generate_conformers(n_samples={n}): {dt:.3f} s -> {len(mols)} molecules returned ({n/dt:.1f} molecules/s through the public API)")
