"""The other ``EDMArchetecture`` parameterisations the reference ships: ``VPODE`` (diffusion/vpode.py:13-78), ``VEODE``
(diffusion/veode.py:13-73), ``EDMDDIM`` (diffusion/edmddim.py:12-84).  Nothing in the reference imports them, and their
``set_sample_steps`` builds N-entry tables where ``EDMArchetecture.sample`` indexes N+1 -- ``sample()`` raises IndexError there and,
faithfully, here; what does work in the reference (construction, preconditioning hooks, ``loss()``/``update()``) works the same.
"""
from typing import Optional, Union

import numpy as np
import torch

from .edm import EDMArchetecture


class _Tabled(EDMArchetecture):
    def _finish_tables(self):
        self.x_weight_s = self.dot_sigma_s / self.sigma_s + self.dot_scale_s / self.scale_s
        self.D_weight_s = self.dot_sigma_s / self.sigma_s * self.scale_s

    def c_skip(self, sigma):
        return torch.ones_like(sigma)

    def loss_weighting(self, sigma):
        return 1 / (sigma ** 2)


class VPODE(_Tabled):
    def __init__(self, nn_diffusion, nn_condition=None, fix_mask=None, loss_weight=None, classifier=None,
                 grad_clip_norm: Optional[float] = None, diffusion_steps: int = 1000, ema_rate: float = 0.995,
                 optim_params: Optional[dict] = None, beta_min: float = 0.1, beta_max: float = 20., eps_s: float = 1e-3,
                 eps_t: float = 1e-5, device: Union[torch.device, str] = "cpu"):
        super().__init__(nn_diffusion, nn_condition, fix_mask, loss_weight, classifier, grad_clip_norm, diffusion_steps,
                         ema_rate, optim_params, device)
        self.beta_min, self.beta_max, self.eps_s, self.eps_t = beta_min, beta_max, eps_s, eps_t
        self.beta_d = beta_max - beta_min

    def _sigma_of_t(self, t):
        return ((0.5 * self.beta_d * t ** 2 + self.beta_min * t).exp() - 1.).sqrt()

    def set_sample_steps(self, N: int):
        self.sample_steps = N
        self.t_s = torch.arange(N, device=self.device) / (N - 1) * (1e-3 - 1) + 1
        self.sigma_s = self._sigma_of_t(self.t_s)
        self.scale_s = 1 / (1 + self.sigma_s ** 2).sqrt()
        self.dot_sigma_s = 0.5 * (self.sigma_s ** 2 + 1) * (self.beta_d * self.t_s + self.beta_min) / self.sigma_s
        self.dot_scale_s = -self.sigma_s / (1 + self.sigma_s ** 2) ** 1.5 * self.dot_sigma_s
        self._finish_tables()

    def c_out(self, sigma):
        return -sigma

    def c_in(self, sigma):
        return 1 / (1 + sigma ** 2).sqrt()

    def c_noise(self, sigma):
        log_scale = (1 / (1 + sigma ** 2).sqrt()).log()
        t = ((self.beta_min ** 2 - 4 * self.beta_d * log_scale).sqrt() - self.beta_min) / self.beta_d
        return ((self.diffusion_steps - 1) * t).long()

    def sample_noise_distribution(self, N):
        return self._sigma_of_t(torch.rand((N, 1), device=self.device) * (1 - self.eps_t) + self.eps_t)


class VEODE(_Tabled):
    def __init__(self, nn_diffusion, nn_condition=None, fix_mask=None, loss_weight=None, classifier=None,
                 grad_clip_norm: Optional[float] = None, diffusion_steps: int = 1000, ema_rate: float = 0.995,
                 optim_params: Optional[dict] = None, sigma_min: float = 0.02, sigma_max: float = 100.,
                 device: Union[torch.device, str] = "cpu"):
        super().__init__(nn_diffusion, nn_condition, fix_mask, loss_weight, classifier, grad_clip_norm, diffusion_steps,
                         ema_rate, optim_params, device)
        self.sigma_min, self.sigma_max = sigma_min, sigma_max

    def set_sample_steps(self, N: int):
        self.sample_steps = N
        self.sigma_s = self.sigma_max * (self.sigma_min / self.sigma_max) ** (torch.arange(N, device=self.device) / (N - 1))
        self.t_s = self.sigma_s ** 2
        self.scale_s = torch.ones_like(self.sigma_s) * 1.0
        self.dot_sigma_s = 1 / (2 * self.sigma_s)
        self.dot_scale_s = torch.zeros_like(self.sigma_s)
        self._finish_tables()

    def c_out(self, sigma):
        return sigma

    def c_in(self, sigma):
        return torch.ones_like(sigma)

    def c_noise(self, sigma):
        return (0.5 * sigma).log()

    def sample_noise_distribution(self, N):
        span = np.log(self.sigma_max / self.sigma_min)
        return (torch.rand((N, 1), device=self.device) * span + np.log(self.sigma_min)).exp()


class EDMDDIM(_Tabled):
    def __init__(self, nn_diffusion, nn_condition=None, fix_mask=None, loss_weight=None, classifier=None,
                 grad_clip_norm: Optional[float] = None, diffusion_steps: int = 1000, ema_rate: float = 0.995,
                 optim_params: Optional[dict] = None, C1: float = 0.001, C2: float = 0.008, j0: float = 8,
                 device: Union[torch.device, str] = "cpu"):
        super().__init__(nn_diffusion, nn_condition, fix_mask, loss_weight, classifier, grad_clip_norm, diffusion_steps,
                         ema_rate, optim_params, device)
        self.C1, self.C2, self.j0 = C1, C2, j0
        self.u = None

    def set_sample_steps(self, N: int):
        self.sample_steps = N
        T = self.diffusion_steps
        bar_alpha = torch.sin(torch.arange(T + 1, device=self.device) / (T * (self.C2 + 1)) * np.pi / 2) ** 2
        ratio = torch.max(bar_alpha[:-1] / bar_alpha[1:], torch.tensor(self.C1, device=self.device))
        self.u = torch.empty_like(bar_alpha[:-1])
        nxt = torch.zeros((), device=self.device)             # u_T = 0:  u_j = sqrt((u_{j+1}^2 + 1) / ratio_j - 1), filled from the top down
        for j in range(T - 1, -1, -1):
            self.u[j] = ((nxt ** 2 + 1) / ratio[j] - 1).sqrt()
            nxt = self.u[j]
        idx = torch.arange(N, device=self.device)
        self.t_s = self.u[torch.floor(self.j0 + (T - 1 - self.j0) / (N - 1) * idx + 0.5).long()]
        self.sigma_s = self.t_s
        self.scale_s = torch.ones_like(self.sigma_s) * 1.0
        self.dot_sigma_s = torch.ones_like(self.sigma_s) * 1.0
        self.dot_scale_s = torch.zeros_like(self.sigma_s)
        self._finish_tables()

    def c_out(self, sigma):
        return -sigma

    def c_in(self, sigma):
        return 1 / (1 + sigma ** 2).sqrt()

    def c_noise(self, sigma):
        return sigma

    def sample_noise_distribution(self, N):
        return self.u[torch.randint(0, self.diffusion_steps, (N, 1), device=self.device)]
