"""Noise schedule table (host side, built once per generator).

Mirrors `PredefinedNoiseSchedule` / `polynomial_schedule` / `clip_noise_schedule`
(equivariant_diffusion.py:9-45,108-134).  The table is float32 and is built with
float32 torch ops in the same order as the reference, so the gamma values (and
hence every per-step scalar) agree bit-for-bit with the reference's table; the
golden KATs in tests/golden/schedule.npz pin this.
"""
from __future__ import annotations

import math

import torch


def gamma_table(timesteps: int, precision: float = 1e-5, power: int = 2) -> torch.Tensor:
    """gamma[0..T] (float32, CPU)."""
    n = timesteps + 1
    grid = torch.linspace(0, n, n)                         # float32, :36
    a2 = (1 - torch.pow(grid / n, power)) ** 2             # :37
    # clip the per-step ratio alpha2_t / alpha2_{t-1} to [0.001, 1]  (:17-22)
    ext = torch.cat((torch.ones(1), a2), dim=0)
    ratio = torch.clip(ext[1:] / ext[:-1], min=0.001, max=1.0)
    a2 = torch.cumprod(ratio, dim=0)
    a2 = (1 - 2 * precision) * a2 + precision              # :41-43
    s2 = 1 - a2                                            # :121
    gamma = -(torch.log(a2) - torch.log(s2))               # :123-130
    return gamma.float().contiguous()


def step_scalars(gamma: torch.Tensor, s_int: int, T: int):
    """Per-step scalars of `sample_p_zs_given_zt` (equivariant_diffusion.py:305-326)
    for the transition t=(s+1)/T -> s/T, as python floats computed in float32.

    Returns (c_z, c_eps, c_noise):  z_s = c_z*z_t - c_eps*eps_hat + c_noise*noise
    with c_z = 1/alpha_ts, c_eps = sigma2_ts/alpha_ts/sigma_t,
    c_noise = sigma_ts*sigma_s/sigma_t.  The reference evaluates the same
    expressions on [B,1] tensors whose rows are identical.
    """
    import torch.nn.functional as F
    # lookup index: round(t*T) with t = s/T as float32 true division (:132-134, :388-391)
    s_arr = torch.tensor([float(s_int)], dtype=torch.float32) / T
    t_arr = (torch.tensor([float(s_int)], dtype=torch.float32) + 1.0) / T
    g_s = gamma[torch.round(s_arr * T).long()]
    g_t = gamma[torch.round(t_arr * T).long()]
    sigma2_ts = 1 - torch.exp(F.softplus(g_s) - F.softplus(g_t))
    alpha_ts = torch.exp(0.5 * (F.logsigmoid(-g_t) - F.logsigmoid(-g_s)))
    sigma_ts = torch.sqrt(sigma2_ts)
    sigma_s = torch.sqrt(torch.sigmoid(g_s))
    sigma_t = torch.sqrt(torch.sigmoid(g_t))
    return (alpha_ts, sigma2_ts / alpha_ts / sigma_t, sigma_ts * sigma_s / sigma_t,
            float(t_arr[0]), float(s_arr[0]))


def lookup(gamma: torch.Tensor, level: float, T: int) -> torch.Tensor:
    """gamma at normalised time `level` in [0,1] -> 1-element float32 tensor."""
    t = torch.tensor([level], dtype=torch.float32)
    return gamma[torch.round(t * T).long()]
