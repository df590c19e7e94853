"""Noise schedules, time discretisations and sampling-step schedules.

Host-side fp32 tables only (length <= a few hundred) -- never on the device hot path.
The registry *names* and call signatures are the plug-in contract of the reference
(cleandiffuser/utils/utils.py:89-233); the arithmetic is restated here so that the
tables are bit-identical to the reference's when evaluated with torch fp32 on CPU.
"""
from typing import Callable, Dict

import numpy as np
import torch

# --------------------------------------------------------------------------- #
# VP noise schedules  alpha(t), sigma(t) = sqrt(1 - alpha^2)                    #
# --------------------------------------------------------------------------- #

_HALF_PI = np.pi / 2.0


def _vp_sigma(alpha: torch.Tensor) -> torch.Tensor:
    return (1.0 - alpha ** 2).sqrt()


def linear_noise_schedule(t_diffusion: torch.Tensor, beta0: float = 0.1, beta1: float = 20.0):
    """log alpha = -(b1-b0)/4 t^2 - b0/2 t   (reference utils.py:99-105)."""
    quad = -(beta1 - beta0) / 4.0 * (t_diffusion ** 2)
    alpha = (quad - beta0 / 2.0 * t_diffusion).exp()
    return alpha, _vp_sigma(alpha)


def inverse_linear_noise_schedule(alpha=None, sigma=None, logSNR=None, beta0: float = 0.1, beta1: float = 20.0):
    """t(lambda) for the linear schedule (reference utils.py:108-119)."""
    assert (logSNR is not None) or (alpha is not None and sigma is not None)
    lam = logSNR if logSNR is not None else (alpha / sigma).log()
    softplus_m2lam = (1 + (-2 * lam).exp()).log()
    return 2 * softplus_m2lam / (beta0 + (beta0 ** 2 + 2 * (beta1 - beta0) * softplus_m2lam))


def cosine_noise_schedule(t_diffusion: torch.Tensor, s: float = 0.008):
    """alpha = cos(pi/2 (clip(t)+s)/(1+s)) / cos(pi/2 s/(1+s))   (reference utils.py:122-126)."""
    alpha = (_HALF_PI * (t_diffusion.clip(0., 0.9946) + s) / (1 + s)).cos() / np.cos(_HALF_PI * s / (1 + s))
    return alpha, _vp_sigma(alpha)


def inverse_cosine_noise_schedule(alpha=None, sigma=None, logSNR=None, s: float = 0.008):
    """t(lambda) for the cosine schedule (reference utils.py:129-141)."""
    assert (logSNR is not None) or (alpha is not None and sigma is not None)
    lam = logSNR if logSNR is not None else (alpha / sigma).log()
    log_alpha = -0.5 * (1 + (-2 * lam).exp()).log()
    inner = (log_alpha + np.log(np.cos(np.pi * s / 2 / (s + 1)))).exp()
    return 2 * (1 + s) / np.pi * torch.arccos(inner) - s


SUPPORTED_NOISE_SCHEDULES: Dict[str, Dict[str, Callable]] = {
    "linear": {"forward": linear_noise_schedule, "reverse": inverse_linear_noise_schedule},
    "cosine": {"forward": cosine_noise_schedule, "reverse": inverse_cosine_noise_schedule},
}


# --------------------------------------------------------------------------- #
# Legacy beta schedules (numpy) used by the old DDPM class (utils.py:77-85)      #
# --------------------------------------------------------------------------- #

def linear_beta_schedule(beta_min: float = 1e-4, beta_max: float = 0.02, T: int = 1000):
    return np.linspace(beta_min, beta_max, T)


def cosine_beta_schedule(s: float = 0.008, T: int = 1000):
    grid = np.arange(T + 1) / T
    f = np.cos((grid + s) / (1 + s) * np.pi / 2.0) ** 2
    bar = f / f[0]
    return (1 - bar[1:] / bar[:-1]).clip(None, 0.999)


# --------------------------------------------------------------------------- #
# Discretisation of [eps, 1] into T diffusion steps                             #
# --------------------------------------------------------------------------- #

def uniform_discretization(T: int = 1000, eps: float = 1e-3):
    return torch.linspace(eps, 1.0, T)


SUPPORTED_DISCRETIZATIONS = {"uniform": uniform_discretization}


# --------------------------------------------------------------------------- #
# Sampling-step schedules: index (discrete) or time (continuous) per sample step #
# Every schedule is "shape(u) over u = linspace(0,1,S+1)" mapped to either       #
# integer indices 0..T-1 or the continuous range [t0, t1].                       #
# --------------------------------------------------------------------------- #

def _unit_grid(sampling_steps: int) -> torch.Tensor:
    return torch.linspace(0, 1, sampling_steps + 1, dtype=torch.float32)


def _shape_quad(u, n):
    return u ** n


def _shape_cat_cos(u, n):
    sign = 2 * (u > 0.5) - 1
    return 0.5 * sign * torch.sin(np.pi * torch.abs(u - 0.5)) ** (1 / n) + 0.5


def _shape_quad_cos(u, n):
    return ((torch.sin(np.pi * (u - 0.5)) + 1) / 2) ** n


def _to_index(T, shaped):
    return ((T - 1) * shaped).to(torch.long)


def _to_range(trange, shaped):
    lo, hi = (1e-3, 1.0) if trange is None else (trange[0], trange[1])
    return (hi - lo) * shaped + lo


def uniform_sampling_step_schedule(T: int = 1000, sampling_steps: int = 10):
    # NOTE: S == T duplicates index 0; kept on purpose.
    return torch.linspace(0, T - 1, sampling_steps + 1, dtype=torch.long)


def uniform_sampling_step_schedule_continuous(trange=None, sampling_steps: int = 10):
    lo, hi = (1e-3, 1.0) if trange is None else (trange[0], trange[1])
    return torch.linspace(lo, hi, sampling_steps + 1, dtype=torch.float32)


def quad_sampling_step_schedule(T: int = 1000, sampling_steps: int = 10, n: int = 1.5):
    return _to_index(T, _shape_quad(_unit_grid(sampling_steps), n))


def quad_sampling_step_schedule_continuous(trange=None, sampling_steps: int = 10, n: int = 1.5):
    return _to_range(trange, _shape_quad(_unit_grid(sampling_steps), n))


def cat_cos_sampling_step_schedule(T: int = 1000, sampling_steps: int = 10, n: int = 2.0):
    return _to_index(T, _shape_cat_cos(_unit_grid(sampling_steps), n))


def cat_cos_sampling_step_schedule_continuous(trange=None, sampling_steps: int = 10, n: int = 2.0):
    return _to_range(trange, _shape_cat_cos(_unit_grid(sampling_steps), n))


def quad_cos_sampling_step_schedule(T: int = 1000, sampling_steps: int = 10, n: int = 2.0):
    return _to_index(T, _shape_quad_cos(_unit_grid(sampling_steps), n))


def quad_cos_sampling_step_schedule_continuous(trange=None, sampling_steps: int = 10, n: int = 2.0):
    return _to_range(trange, _shape_quad_cos(_unit_grid(sampling_steps), n))


SUPPORTED_SAMPLING_STEP_SCHEDULE = {
    "uniform": uniform_sampling_step_schedule,
    "uniform_continuous": uniform_sampling_step_schedule_continuous,
    "quad": quad_sampling_step_schedule,
    "quad_continuous": quad_sampling_step_schedule_continuous,
    "cat_cos": cat_cos_sampling_step_schedule,
    "cat_cos_continuous": cat_cos_sampling_step_schedule_continuous,
    "quad_cos": quad_cos_sampling_step_schedule,
    "quad_cos_continuous": quad_cos_sampling_step_schedule_continuous,
}
