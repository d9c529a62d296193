"""Oracle: variance-preserving ancestral sampler (test infrastructure - see oracle/__init__.py).

Restates `EquivariantDiffusion` forward / inpaint / merge_fragments
(equivariant_diffusion.py:137-607) on torch-CPU fp32 with the same op order and the
same RNG draw order (x-draw [B,N,3] then h-draw [B,N,8] per noise sample).
`noise_fn(shape) -> tensor` may be injected to replay a recorded noise tape.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .egnn_oracle import egnn_dynamics, masked_mean_removal


def gamma_schedule(timesteps: int, precision: float = 1e-5, power: int = 2) -> torch.Tensor:
    """equivariant_diffusion.py:9-45,113-130: float32 lookup table gamma[0..T]."""
    steps = timesteps + 1
    u = torch.linspace(0, steps, steps)
    alphas2 = (1 - torch.pow(u / steps, power)) ** 2
    padded = torch.cat((torch.ones(1), alphas2), dim=0)
    step_ratio = torch.clip(padded[1:] / padded[:-1], min=0.001, max=1.0)
    alphas2 = torch.cumprod(step_ratio, dim=0)
    alphas2 = (1 - 2 * precision) * alphas2 + precision
    sigmas2 = 1 - alphas2
    return (-(torch.log(alphas2) - torch.log(sigmas2))).float()


class SamplerOracle:
    def __init__(self, sd: Dict[str, torch.Tensor], timesteps: int, precision: float = 1e-5,
                 noise_fn: Optional[Callable] = None, n_blocks: int = 9,
                 n_classes: int = 8, norm_values=(1.0, 9.0)):
        self.sd = sd
        self.T = timesteps
        self.gamma = gamma_schedule(timesteps, precision)
        self.noise_fn = noise_fn or (lambda shape: torch.randn(shape))
        self.n_blocks = n_blocks
        self.n_classes = n_classes
        self.norm_values = norm_values
        self.trace: Optional[List[torch.Tensor]] = None   # every z_s when recording

    # -- schedule algebra (:132-134, :190-247)
    def g(self, t: torch.Tensor) -> torch.Tensor:
        return self.gamma[torch.round(t * self.T).long()]

    @staticmethod
    def _inflate(a: torch.Tensor) -> torch.Tensor:
        return a.view(a.size(0), 1, 1)

    def sigma(self, gamma):
        return self._inflate(torch.sqrt(torch.sigmoid(gamma)))

    def alpha(self, gamma):
        return self._inflate(torch.sqrt(torch.sigmoid(-gamma)))

    # -- network call (:176-188)
    def phi(self, z, t, node_mask, edge_mask, context):
        return egnn_dynamics(self.sd, t, z, node_mask, edge_mask, context, self.n_blocks)

    # -- noise (:56-76, :341-363): x-draw, mask, centre; then h-draw, mask
    def draw(self, node_mask: torch.Tensor) -> torch.Tensor:
        B, N, _ = node_mask.shape
        ex = self.noise_fn((B, N, 3)) * node_mask
        ex = masked_mean_removal(ex, node_mask)
        eh = self.noise_fn((B, N, self.n_classes)) * node_mask
        return torch.cat([ex, eh], dim=2)

    # -- one ancestral step (:295-339)
    def step(self, s, t, zt, node_mask, edge_mask, context):
        g_s, g_t = self.g(s), self.g(t)
        sigma2_ts = self._inflate(1 - torch.exp(F.softplus(g_s) - F.softplus(g_t)))
        alpha_ts = self._inflate(torch.exp(0.5 * (F.logsigmoid(-g_t) - F.logsigmoid(-g_s))))
        sigma_ts = torch.sqrt(sigma2_ts)
        sigma_s, sigma_t = self.sigma(g_s), self.sigma(g_t)
        eps = self.phi(zt, t, node_mask, edge_mask, context)
        mu = zt / alpha_ts - (sigma2_ts / alpha_ts / sigma_t) * eps
        zs = mu + (sigma_ts * sigma_s / sigma_t) * self.draw(node_mask)
        zs = torch.cat([masked_mean_removal(zs[:, :, :3], node_mask), zs[:, :, 3:]], dim=2)
        if self.trace is not None:
            self.trace.append(zs.clone())
        return zs

    # -- final decode (:261-285).  NB argmax over 7 of the 8 class channels (z0[:,:,3:-1]).
    def decode(self, z0, node_mask, edge_mask, context):
        zeros = torch.zeros((z0.size(0), 1))
        g0 = self.g(zeros)
        sigma_x = torch.exp(-(-0.5 * g0)).unsqueeze(1)          # snr(-gamma/2) = exp(gamma/2)
        eps = self.phi(z0, zeros, node_mask, edge_mask, context)
        mu = 1.0 / self.alpha(g0) * (z0 - self.sigma(g0) * eps)
        xh = mu + sigma_x * self.draw(node_mask)
        x = xh[:, :, :3] * self.norm_values[0]
        h_cat = z0[:, :, 3:-1] * self.norm_values[1] * node_mask
        h = F.one_hot(torch.argmax(h_cat, dim=2), self.n_classes) * node_mask
        return x, h

    def _times(self, s_int: int, B: int):
        s = torch.full([B, 1], fill_value=s_int)                 # int64, :388
        t = s + 1.0
        return s / self.T, t / self.T

    # -- :365-421
    def forward(self, node_mask, edge_mask, context, resample_steps: int = 0):
        B = node_mask.size(0)
        z = self.draw(node_mask)
        for s_int in range(self.T - 1, -1, -1):
            s, t = self._times(s_int, B)
            for _ in range(resample_steps + 1):
                z = self.step(s, t, z, node_mask, edge_mask, context)
        return self.decode(z, node_mask, edge_mask, context)

    # -- :79-105
    @staticmethod
    def _align_fragment(z_known_noised, z_gen, fixed_mask):
        cnt = fixed_mask.sum(dim=1, keepdim=True)
        com_gen = (z_gen[:, :, :3] * fixed_mask).sum(dim=1, keepdim=True) / cnt
        com_known = (z_known_noised[:, :, :3] * fixed_mask).sum(dim=1, keepdim=True) / cnt
        out = z_known_noised.clone()
        out[:, :, :3] = z_known_noised[:, :, :3] + (com_gen - com_known) * fixed_mask
        return out

    def _blend_known(self, z, s, z_known, fixed_mask, node_mask, blend):
        g_s = self.g(s)
        noised = self.alpha(g_s) * z_known + self.sigma(g_s) * self.draw(node_mask)
        noised = self._align_fragment(noised, z, fixed_mask)
        return blend * noised * fixed_mask + (1 - blend) * z * fixed_mask + z * (1 - fixed_mask)

    # -- :423-513
    def inpaint(self, node_mask, edge_mask, context, z_known, fixed_mask,
                resample_steps: int = 1, blend_power: int = 3):
        resample_steps = max(1, resample_steps)
        B = node_mask.size(0)
        z = self.draw(node_mask)
        for s_int in range(self.T - 1, -1, -1):
            s, t = self._times(s_int, B)
            blend = torch.pow((1 - s), blend_power).view(B, 1, 1)
            for _ in range(resample_steps):
                z = self.step(s, t, z, node_mask, edge_mask, context)
                z = self._blend_known(z, s, z_known, fixed_mask, node_mask, blend)
            z = self.step(s, t, z, node_mask, edge_mask, context)   # harmonisation pass
        return self.decode(z, node_mask, edge_mask, context)

    # -- :515-607
    def merge_fragments(self, node_mask, edge_mask, fixed_mask, context, z_known,
                        diffusion_level: int = 50, resample_steps: int = 1, blend_power: int = 3):
        resample_steps = max(1, resample_steps)
        B = node_mask.size(0)
        s0 = torch.full([B, 1], fill_value=diffusion_level) / self.T
        g0 = self.g(s0)                                   # IndexError if level > T (quirk H5)
        z = self.alpha(g0) * z_known + self.sigma(g0) * self.draw(node_mask)
        for s_int in range(self.T - 1, -1, -1):
            if s_int > diffusion_level:
                continue
            s, t = self._times(s_int, B)
            blend = torch.pow((1 - s), blend_power).view(B, 1, 1)
            for _ in range(resample_steps):
                z = self.step(s, t, z, node_mask, edge_mask, context)
                z = self._blend_known(z, s, z_known, fixed_mask, node_mask, blend)
        return self.decode(z, node_mask, edge_mask, context)
