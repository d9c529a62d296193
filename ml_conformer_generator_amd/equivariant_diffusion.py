"""Variance-preserving ancestral sampler on the HIP path.

Host-side mirror of `EquivariantDiffusion` (equivariant_diffusion.py:137-607): same
method names, argument order and RNG draw order (one `randn[B,N,3]` then one
`randn[B,N,8]` per noise sample, on the model's device).  Per-step scalars are
evaluated once on the host with the reference's fp32 expressions; everything that
touches a [B,N,*] tensor runs in libmlconfgen_hip.so (one network call + one fused
update launch per step).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

import torch
import torch.nn.functional as F

from . import _lib
from .config import N_ATOM_CLASSES, NORM_VALUES
from .egnn import BatchPlan, EGNNDynamics, sizes_from_node_mask
from .schedule import gamma_table, step_scalars


class PredefinedNoiseSchedule(torch.nn.Module):
    """Lookup table gamma[0..T] (equivariant_diffusion.py:108-134); kept on the host."""

    def __init__(self, timesteps: int, precision: float, power: int = 2):
        super().__init__()
        self.timesteps = timesteps
        self.gamma = torch.nn.Parameter(gamma_table(timesteps, precision, power), requires_grad=False)

    def forward(self, t: torch.Tensor) -> torch.Tensor:
        return self.gamma[torch.round(t.cpu() * self.timesteps).long()]


class EquivariantDiffusion(torch.nn.Module):
    def __init__(self, dynamics: EGNNDynamics, in_node_nf: int = N_ATOM_CLASSES, n_dims: int = 3,
                 timesteps: int = 1000, noise_precision: float = 1e-4,
                 norm_values: Tuple[float, float] = NORM_VALUES):
        super().__init__()
        self.gamma = PredefinedNoiseSchedule(timesteps=timesteps, precision=noise_precision)
        self.dynamics = dynamics
        self.in_node_nf, self.n_dims, self.num_classes = in_node_nf, n_dims, in_node_nf
        self.T = timesteps
        self.time_steps = torch.flip(torch.arange(0, timesteps), dims=[0])
        self.norm_values = norm_values
        self.noise_fn: Optional[Callable] = None     # tests inject a recorded noise tape here
        self.trace: Optional[List[torch.Tensor]] = None
        self._scalar_cache = {}

    @property
    def device(self) -> torch.device:
        return self.dynamics.device

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, state_dict, strict: bool = True):  # noqa: D401 - reference signature
        """Loads a reference `EquivariantDiffusion.state_dict()` (keys `gamma.gamma`,
        `dynamics.egnn.*`; conformer_generator.py:90-95)."""
        if strict and "gamma.gamma" not in state_dict:
            raise RuntimeError('Missing key(s) in state_dict: "gamma.gamma"')
        self.dynamics.load_reference_state_dict(state_dict, prefix="dynamics.egnn.")
        if "gamma.gamma" in state_dict and state_dict["gamma.gamma"].numel() == self.gamma.gamma.numel():
            self.gamma.gamma.data.copy_(state_dict["gamma.gamma"].to(torch.float32).cpu())
        return None

    # ------------------------------------------------------------------ schedule algebra (host, fp32)
    def _g(self, level_int: int) -> torch.Tensor:
        """gamma at t = level/T, with the reference's float32 index arithmetic (:132-134,:388-391)."""
        t = torch.full([1, 1], fill_value=level_int) / self.T
        return self.gamma(t)

    def _step_scalars(self, s_int: int):
        """(alpha_ts, c_eps, c_noise) of sample_p_zs_given_zt (:305-326) as python floats (fp32 values);
        cached per (schedule, step): they depend on nothing else."""
        key = (id(self.gamma), self.T, s_int)
        hit = self._scalar_cache.get(key)
        if hit is None:
            a, c_eps, c_noise, _, _ = step_scalars(self.gamma.gamma.detach(), s_int, self.T)
            hit = (float(a), float(c_eps), float(c_noise))
            self._scalar_cache[key] = hit
        return hit

    def _alpha_sigma(self, level_int: int):
        g = self._g(level_int)
        return float(torch.sqrt(torch.sigmoid(-g))), float(torch.sqrt(torch.sigmoid(g)))

    # ------------------------------------------------------------------ device helpers
    def _randn(self, shape):
        if self.noise_fn is not None:
            return self.noise_fn(shape).to(self.device, torch.float32).contiguous()
        return torch.randn(shape, device=self.device)

    def _draw(self, B: int, N: int):
        """Raw draws in the reference's order: x block first, then h block (:347-361)."""
        rx = self._randn((B, N, self.n_dims))
        rh = self._randn((B, N, self.in_node_nf))
        return rx, rh

    class _Run:
        """State of one sampling call: plan, device context, time table, scratch.

        The latent `z`, the network output and the context live in buffers OWNED BY THE CACHED PLAN (so that the captured
        HIP graph is replayed, not re-captured).  A plan - and therefore a sampling run over a given batch of sizes - is
        single-stream and non-reentrant, like the reference's sampler (one Python thread, one stream): two runs that
        resolve to the same plan must not overlap in time (threads, streams, or two EquivariantDiffusion objects over one
        dynamics).  What a run RETURNS (`x`, `h` of `_decode`, the tensor of `sample_combined_position_feature_noise`) is
        always a fresh tensor and is never overwritten by a later run; the in-place latent handed to `trace` is cloned."""

        def __init__(self, model: "EquivariantDiffusion", node_mask, context, edge_mask=None):
            dev = model.device
            self.B, self.N = int(node_mask.shape[0]), int(node_mask.shape[1])
            self.plan: BatchPlan = model.dynamics.plan(model.dynamics.sizes_for(node_mask), self.N)
            model.dynamics.check_edge_mask(self.plan, edge_mask, node_mask)      # a non-canonical edge mask is refused, not ignored
            # Latent, network output and context live in buffers owned by the (cached) plan: the denoiser call is a HIP
            # graph keyed by these addresses, so a second sampling run over the same batch shape replays the captured
            # graph instead of re-capturing it (or, from the third run on, paying three staging copies per call).
            sc = getattr(self.plan, "_sampler_scratch", None)
            if sc is None:
                sc = {k: torch.empty((self.B, self.N, w), device=dev, dtype=torch.float32) for k, w in (("z", 11), ("eps", 11), ("ctx", 3))}
                self.plan._sampler_scratch = sc
            self.context = sc["ctx"]
            self.context.copy_(context.to(dev, torch.float32).reshape(self.B, self.N, 3))
            self.z_buf = sc["z"]
            T = model.T
            # t value of every level 0..T, as float32(level)/T  (:388-391)
            levels = torch.arange(0, T + 1).to(torch.float32) / T
            self.t_table = levels.unsqueeze(1).repeat(1, self.B).to(dev).contiguous()   # [T+1, B]
            self.eps_hat = sc["eps"]
            self.stream = _lib.current_stream_ptr(dev)

    def sample_combined_position_feature_noise(self, n_samples: int, n_nodes: int, node_mask) -> torch.Tensor:
        """Mean-centred x noise + plain h noise, masked (:341-363)."""
        run = self._Run(self, node_mask, torch.zeros(n_samples, n_nodes, 3))
        return self._noise(run)

    def _noise(self, run: "_Run", out: Optional[torch.Tensor] = None) -> torch.Tensor:
        rx, rh = self._draw(run.B, run.N)
        eps = out if out is not None else torch.empty((run.B, run.N, 11), device=self.device, dtype=torch.float32)
        _lib.check(_lib.lib().mcg_sampler_noise(run.plan.handle, _lib.dptr(rx), _lib.dptr(rh), _lib.dptr(eps),
                                                run.stream), "mcg_sampler_noise")
        return eps

    def phi(self, x, t, node_mask, edge_mask, context):
        """Denoising pass (:176-188)."""
        return self.dynamics(t, x, node_mask, edge_mask, context)

    def _step(self, run: "_Run", z: torch.Tensor, s_int: int) -> torch.Tensor:
        """In-place z_t -> z_s for t = (s+1)/T (:295-339)."""
        alpha_ts, c_eps, c_noise = self._step_scalars(s_int)
        rx, rh = self._draw(run.B, run.N)
        _lib.check(_lib.lib().mcg_sampler_step(
            self.dynamics.handle, run.plan.handle, _lib.dptr(z), _lib.dptr(run.context),
            run.t_table[s_int + 1].data_ptr(), _lib.dptr(rx), _lib.dptr(rh), alpha_ts, c_eps, c_noise,
            _lib.dptr(run.eps_hat), run.stream), "mcg_sampler_step")
        if self.trace is not None:
            self.trace.append(z.clone())
        return z

    def _decode(self, run: "_Run", z0: torch.Tensor):
        """x, h = sample_p_xh_given_z0 (:261-285)."""
        g0 = self._g(0)
        sigma_x = float(torch.exp(-(-0.5 * g0)))
        alpha0, sigma0 = float(torch.sqrt(torch.sigmoid(-g0))), float(torch.sqrt(torch.sigmoid(g0)))
        inv_alpha0 = float(1.0 / torch.sqrt(torch.sigmoid(-g0)))
        rx, _rh = self._draw(run.B, run.N)      # the h block is drawn (RNG parity) but unused (:277-285)
        x = torch.empty((run.B, run.N, 3), device=self.device, dtype=torch.float32)
        h = torch.empty((run.B, run.N, self.num_classes), device=self.device, dtype=torch.float32)
        _lib.check(_lib.lib().mcg_sampler_decode(
            self.dynamics.handle, run.plan.handle, _lib.dptr(z0), _lib.dptr(run.context), run.t_table[0].data_ptr(),
            _lib.dptr(rx), inv_alpha0, sigma0, sigma_x, float(self.norm_values[0]), float(self.norm_values[1]),
            _lib.dptr(run.eps_hat), _lib.dptr(x), _lib.dptr(h), run.stream), "mcg_sampler_decode")
        return x, h

    def _blend(self, run: "_Run", z, z_known, fixed_mask, s_int: int, blend_power: int, mode: int):
        alpha_s, sigma_s = self._alpha_sigma(s_int)
        s_arr = torch.full([1, 1], fill_value=s_int) / self.T
        blend = float(torch.pow((1 - s_arr), blend_power)) if mode == 1 else 0.0
        rx, rh = self._draw(run.B, run.N)
        _lib.check(_lib.lib().mcg_sampler_blend(
            run.plan.handle, _lib.dptr(z), _lib.dptr(z_known), _lib.dptr(fixed_mask) if fixed_mask is not None else None,
            _lib.dptr(rx), _lib.dptr(rh), alpha_s, sigma_s, blend, mode, run.stream), "mcg_sampler_blend")
        return z

    # ------------------------------------------------------------------ public sampling entry points
    @torch.no_grad()
    def forward(self, node_mask, edge_mask, context, resample_steps: int = 0):
        """Draw samples (:365-421): T*(1+resample_steps) network calls + 1 decode call."""
        run = self._Run(self, node_mask, context, edge_mask)
        z = self._noise(run, out=run.z_buf)
        for s_int in range(self.T - 1, -1, -1):
            for _ in range(resample_steps + 1):
                z = self._step(run, z, s_int)
        return self._decode(run, z)

    @torch.no_grad()
    def inpaint(self, node_mask, edge_mask, context, z_known, fixed_mask, resample_steps: int = 1,
                blend_power: int = 3):
        """Sampling with a fixed fragment blended back in every step (:423-513)."""
        resample_steps = max(1, resample_steps)
        run = self._Run(self, node_mask, context, edge_mask)
        zk = z_known.to(self.device, torch.float32).contiguous()
        fm = fixed_mask.to(self.device, torch.float32).contiguous()
        z = self._noise(run, out=run.z_buf)
        for s_int in range(self.T - 1, -1, -1):
            for _ in range(resample_steps):
                z = self._step(run, z, s_int)
                z = self._blend(run, z, zk, fm, s_int, blend_power, 1)
            z = self._step(run, z, s_int)       # harmonisation pass (:495-503)
        return self._decode(run, z)

    @torch.no_grad()
    def merge_fragments(self, node_mask, edge_mask, fixed_mask, context, z_known, diffusion_level: int = 50,
                        resample_steps: int = 1, blend_power: int = 3):
        """Forward-diffuse the assembled molecule to `diffusion_level`, denoise with the fixed
        fragment blended in (:515-607).  Like the reference, diffusion_level > T raises IndexError."""
        resample_steps = max(1, resample_steps)
        if diffusion_level > self.T or diffusion_level < 0:
            raise IndexError(f"index {diffusion_level} is out of bounds for dimension 0 with size {self.T + 1}")
        run = self._Run(self, node_mask, context, edge_mask)
        zk = z_known.to(self.device, torch.float32).contiguous()
        fm = fixed_mask.to(self.device, torch.float32).contiguous()
        z = self._blend(run, run.z_buf, zk, None, diffusion_level, blend_power, 0)
        for s_int in range(self.T - 1, -1, -1):
            if s_int > diffusion_level:
                continue
            for _ in range(resample_steps):
                z = self._step(run, z, s_int)
                z = self._blend(run, z, zk, fm, s_int, blend_power, 1)
        return self._decode(run, z)
