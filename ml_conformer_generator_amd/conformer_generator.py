"""`MLConformerGenerator` - drop-in for the reference class on the denoising hot path.

Same constructor kwargs, method names, argument order and errors as
`mlconfgen.MLConformerGenerator` (conformer_generator.py:19-399).  Differences, all
forced by the scope of this build:
  * `device` must be a ROCm GPU (default cuda:0): there is no CPU path;
  * `edm_weights` / `adj_mat_seer_weights` may also be an in-memory state dict;
  * without RDKit the RDKit-owned stages are replaced by the native hand-off of
    `handoff.py` and results are `GeneratedMolecule` records instead of `Chem.Mol`.
"""
from __future__ import annotations

from typing import List, Optional, Union

import torch

from . import _lib
from .adj_mat_seer import AdjMatSeer
from .config import (ATOM_DECODER, CONTEXT_NORMS, DIMENSION, MAX_N_NODES, MIN_N_NODES, NOISE_PRECISION,
                     NUM_BOND_TYPES)
from .egnn import EGNNDynamics
from .equivariant_diffusion import EquivariantDiffusion, PredefinedNoiseSchedule
from .handoff import (GeneratedMolecule, assemble_molecules, prepare_adj_mat_seer_input_hip,
                      prepare_adj_mat_seer_input_native)
from .mol_utils import (get_context_shape, ifm_get_xh_from_fragment, ifm_prepare_fragments_for_merge,
                        ifm_prepare_gen_fragment_context, inverse_coord_transform, parse_molblock_heavy_atoms,
                        prepare_edm_input, prepare_fragment)

try:  # RDKit is optional on this path
    from rdkit import Chem  # type: ignore
    HAVE_RDKIT = True
except Exception:  # noqa: BLE001
    Chem = None
    HAVE_RDKIT = False


def _load_state_dict(src, device) -> dict:
    if isinstance(src, dict):
        return src["state_dict"] if "state_dict" in src else src
    return torch.load(src, map_location="cpu")["state_dict"]      # conformer_generator.py:90-102


class MLConformerGenerator(torch.nn.Module):
    def __init__(self, diffusion_steps: int = 100, device: Optional[torch.device] = None,
                 dimension: int = DIMENSION, num_bond_types: int = NUM_BOND_TYPES,
                 min_n_nodes: int = MIN_N_NODES, max_n_nodes: int = MAX_N_NODES,
                 context_norms: dict = CONTEXT_NORMS, atom_decoder: dict = ATOM_DECODER,
                 edm_weights: Union[str, dict] = "./edm_moi_chembl_15_39.pt",
                 adj_mat_seer_weights: Union[str, dict] = "./adj_mat_seer_chembl_15_39.pt",
                 compute_dtype: str = "f32"):
        super().__init__()
        _lib.lib()       # fail loudly if the HIP library is not built
        device = torch.device("cuda:0" if device is None else device)
        if device.type != "cuda":
            raise ValueError("ml_conformer_generator_amd runs the hot path on an MI355X only; "
                             f"device={device} is not supported (no CPU fallback)")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.device = device
        self.dimension = dimension
        self.context_norms = {k: torch.tensor(v) for k, v in context_norms.items()}
        self.atom_decoder = atom_decoder
        self.min_n_nodes, self.max_n_nodes = min_n_nodes, max_n_nodes

        net_dynamics = EGNNDynamics(in_node_nf=9, context_node_nf=3, hidden_nf=420, device=device)
        generative_model = EquivariantDiffusion(dynamics=net_dynamics, in_node_nf=8, timesteps=1000,
                                                noise_precision=NOISE_PRECISION)
        adj_mat_seer = AdjMatSeer(dimension=dimension, n_hidden=2048, embedding_dim=64, num_embeddings=36,
                                  num_bond_types=num_bond_types, device=device)
        generative_model.load_state_dict(_load_state_dict(edm_weights, device))
        net_dynamics.set_precision(compute_dtype)      # opt-in: "bf16" reduced-precision operands, "f32x6" split-operand fp32
        adj_mat_seer.load_state_dict(_load_state_dict(adj_mat_seer_weights, device))
        # re-wire the schedule to the requested number of steps (conformer_generator.py:105-113)
        generative_model.gamma = PredefinedNoiseSchedule(timesteps=diffusion_steps, precision=NOISE_PRECISION)
        generative_model.time_steps = torch.flip(torch.arange(0, diffusion_steps), dims=[0])
        generative_model.T = diffusion_steps
        self.generative_model = generative_model
        self.adj_mat_seer = adj_mat_seer
        self.last_batch = None       # tensors of the most recent generation (x, h, n_nodes, bond)

    # ------------------------------------------------------------------ EDM stage
    @torch.no_grad()
    def edm_tensors(self, reference_context: torch.Tensor, n_samples: int = 100, max_n_nodes: int = 32,
                    min_n_nodes: int = 25, resample_steps: int = 0, fixed_fragment=None,
                    inertial_fragment_matching: bool = True, blend_power: int = 3, ifm_diffusion_level: int = 50):
        """Tensor form of `edm_samples` (conformer_generator.py:125-266): x[B,N,3], h[B,N,8], node_mask."""
        min_n_nodes = max(min_n_nodes, self.min_n_nodes)            # :156-160
        max_n_nodes = min(max_n_nodes, self.max_n_nodes)
        node_mask, edge_mask, batch_context = prepare_edm_input(
            n_samples=n_samples, reference_context=reference_context, context_norms=self.context_norms,
            min_n_nodes=min_n_nodes, max_n_nodes=max_n_nodes, device=self.device)
        gm = self.generative_model
        if fixed_fragment is None:
            x, h = gm(node_mask, edge_mask, batch_context, resample_steps)
        elif inertial_fragment_matching:
            # generate the complementary fragments separately, then merge (:179-240)
            n_nodes = torch.sum(node_mask, dim=1).to(torch.long)
            ff_x, ff_h = ifm_get_xh_from_fragment(fixed_fragment, self.device)
            f_nm, f_em, f_ctx, shift, rotation = ifm_prepare_gen_fragment_context(
                fixed_fragment_x=ff_x, reference_context=reference_context, n_nodes=n_nodes,
                context_norms=self.context_norms, max_n_nodes=max_n_nodes, min_n_nodes=min_n_nodes,
                device=self.device)
            xg, hg = gm(f_nm, f_em, f_ctx, resample_steps)
            xg = inverse_coord_transform(coord=xg, shift=shift, rotation=rotation)
            z_known, fixed_mask = ifm_prepare_fragments_for_merge(
                fixed_fragment_x=ff_x, fixed_fragment_h=ff_h.to(torch.float32), gen_fragments_x=xg,
                gen_fragments_h=hg, device=self.device, max_n_nodes=max_n_nodes)
            x, h = gm.merge_fragments(node_mask=node_mask, edge_mask=edge_mask, fixed_mask=fixed_mask,
                                      context=batch_context, z_known=z_known, diffusion_level=ifm_diffusion_level,
                                      resample_steps=resample_steps, blend_power=blend_power)
        else:
            z_known, fixed_mask = prepare_fragment(n_samples=n_samples, fixed_fragment=fixed_fragment,
                                                   max_n_nodes=max_n_nodes, min_n_nodes=min_n_nodes,
                                                   device=self.device)
            x, h = gm.inpaint(node_mask, edge_mask, batch_context, z_known, fixed_mask, resample_steps, blend_power)
        return x, h, node_mask

    @torch.no_grad()
    def edm_samples(self, reference_context: torch.Tensor, n_samples: int = 100, max_n_nodes: int = 32,
                    min_n_nodes: int = 25, resample_steps: int = 0, fixed_fragment=None,
                    inertial_fragment_matching: bool = True, blend_power: int = 3, ifm_diffusion_level: int = 50):
        """Samples without bonds: RDKit mols when RDKit is present, else GeneratedMolecule records."""
        x, h, node_mask = self.edm_tensors(reference_context, n_samples, max_n_nodes, min_n_nodes, resample_steps,
                                           fixed_fragment, inertial_fragment_matching, blend_power,
                                           ifm_diffusion_level)
        if HAVE_RDKIT:
            from .rdkit_glue import samples_to_rdkit_mol
            return samples_to_rdkit_mol(positions=x, one_hot=h, node_mask=node_mask, atom_decoder=self.atom_decoder)
        n_nodes = node_mask.sum(1).reshape(-1).to(torch.long)
        el, _, _ = prepare_adj_mat_seer_input_native(x, h, n_nodes, self.dimension)
        no_bonds = torch.zeros(x.shape[0], self.dimension, self.dimension, dtype=torch.int8)
        return assemble_molecules(x, el, no_bonds, n_nodes)

    # ------------------------------------------------------------------ full pipeline
    def _reference_context(self, reference_conformer, reference_context, n_atoms):
        if reference_conformer is not None and reference_conformer is not False:
            if HAVE_RDKIT and not isinstance(reference_conformer, (str, torch.Tensor)):
                mol = Chem.RemoveHs(reference_conformer)                       # :302-307
                ref_coord = torch.tensor(mol.GetConformer().GetPositions(), dtype=torch.float32)
            elif isinstance(reference_conformer, str):
                ref_coord, _ = parse_molblock_heavy_atoms(reference_conformer)
            else:
                ref_coord = torch.as_tensor(reference_conformer, dtype=torch.float32)
            ref_n_atoms = int(ref_coord.shape[0])
            ref_coord = ref_coord - torch.mean(ref_coord, dim=0)              # :310-311
            ref_context, _ = get_context_shape(ref_coord)
            return ref_context, ref_n_atoms
        if reference_context is not None:
            if not n_atoms:
                raise ValueError(
                    "Reference Number of Atoms should be provided, when generating samples using context.")
            return reference_context, n_atoms
        raise ValueError(
            "Either a reference RDkit Mol object or context as torch.Tensor should be provided for generation.")

    @torch.no_grad()
    def generate_conformers(self, reference_conformer=None, n_samples: int = 10, variance: int = 2,
                            reference_context: torch.Tensor = None, n_atoms: int = None,
                            optimise_geometry: bool = True, resample_steps: int = 0, fixed_fragment=None,
                            inertial_fragment_matching: bool = True, blend_power: int = 3,
                            ifm_diffusion_level: int = 50) -> List:
        """Generate molecules from a reference shape (conformer_generator.py:268-368).
        Returns the VALID molecules only (invalid ones are dropped, as in the reference)."""
        ref_context, ref_n_atoms = self._reference_context(reference_conformer, reference_context, n_atoms)
        x, h, node_mask = self.edm_tensors(
            reference_context=ref_context, n_samples=n_samples, min_n_nodes=ref_n_atoms - variance,
            max_n_nodes=ref_n_atoms + variance, resample_steps=resample_steps, fixed_fragment=fixed_fragment,
            inertial_fragment_matching=inertial_fragment_matching, blend_power=blend_power,
            ifm_diffusion_level=ifm_diffusion_level)
        n_nodes = node_mask.sum(1).reshape(-1).to(torch.long)
        if HAVE_RDKIT:
            from .rdkit_glue import finish_with_rdkit
            return finish_with_rdkit(self, x, h, node_mask, optimise_geometry)
        el, dm, am = prepare_adj_mat_seer_input_hip(x, h, n_nodes, self.dimension)
        bond = self.adj_mat_seer.bond_orders(el, dm, am)
        self.last_batch = dict(x=x, h=h, n_nodes=n_nodes, elements=el, bond=bond)
        mols = assemble_molecules(x, el, bond, n_nodes)           # single D2H
        return [m for m in mols if m.valid]

    @torch.no_grad()
    def forward(self, reference_conformer=None, n_samples: int = 10, variance: int = 2,
                reference_context: torch.Tensor = None, n_atoms: int = None, optimise_geometry: bool = True,
                resample_steps: int = 0, fixed_fragment=None, inertial_fragment_matching: bool = True,
                blend_power: int = 3, ifm_diffusion_level: int = 50) -> List:
        return self.generate_conformers(reference_conformer, n_samples, variance, reference_context, n_atoms,
                                        optimise_geometry, resample_steps, fixed_fragment,
                                        inertial_fragment_matching, blend_power, ifm_diffusion_level)
