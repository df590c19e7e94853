"""Legacy ``EDM`` solver (``cleandiffuser.diffusion.edm.EDM``; Karras et al. preconditioning with an explicit per-step table).

Contract: reference diffusion/edm.py:15-57 (``EDMArchetecture`` ctor + per-step tables ``sigma_s, t_s, scale_s,
x_weight_s, D_weight_s``), :77-82 (``D``), :87-113 (training), :118-160 (``dot_x``: CFG on D, classifier shift, masked
derivative), :162-275 (``sample``: Euler / Heun with the ``sigma_{i+1} > 0.005`` guard), :277-352 (``sample_x``: extra Euler
steps at the last level), :355-426 (``EDM``: ``sigma_i = (smax^(1/rho) + i/N (smin^(1/rho) - smax^(1/rho)))^rho``, scale 1).

Execution: for the ``EDM`` class proper the loop is one native call on a ROCm device (``build_legacy_edm_plan`` -> step
kinds 5/6); subclasses with their own tables, CPU tensors and classifier guidance use the PyTorch loop below.
"""
from typing import Optional, Union

import numpy as np
import torch
import torch.nn as nn

from ..nn_condition import BaseNNCondition
from ..nn_diffusion import BaseNNDiffusion
from ..utils import at_least_ndim
from .basic import DiffusionModel
from .diffusionsde import _NoiseFeed


class EDMArchetecture(DiffusionModel):
    def __init__(self, nn_diffusion: BaseNNDiffusion, nn_condition: Optional[BaseNNCondition] = None,
                 fix_mask=None, loss_weight=None, classifier=None, grad_clip_norm: Optional[float] = None,
                 diffusion_steps: int = 1000, ema_rate: float = 0.995, optim_params: Optional[dict] = None,
                 device: Union[torch.device, str] = "cpu"):
        super().__init__(nn_diffusion, nn_condition, fix_mask, loss_weight, classifier, grad_clip_norm, diffusion_steps,
                         ema_rate, optim_params, device)
        self.dot_scale_s = self.dot_sigma_s = self.scale_s = self.t_s = self.sigma_s = None
        self.x_weight_s = self.D_weight_s = None
        self.sample_steps = None

    # hooks of a concrete parameterisation
    def set_sample_steps(self, N: int):
        raise NotImplementedError

    def c_skip(self, sigma):
        raise NotImplementedError

    def c_out(self, sigma):
        raise NotImplementedError

    def c_in(self, sigma):
        raise NotImplementedError

    def c_noise(self, sigma):
        raise NotImplementedError

    def loss_weighting(self, sigma):
        raise NotImplementedError

    def sample_noise_distribution(self, N):
        raise NotImplementedError

    def sample_scale_distribution(self, N):
        raise NotImplementedError

    def D(self, x, sigma, condition=None, use_ema=False):
        net = (self.model_ema if use_ema else self.model)["diffusion"]
        noise = at_least_ndim(self.c_noise(sigma).squeeze(), 1)
        return self.c_skip(sigma) * x + self.c_out(sigma) * net(self.c_in(sigma) * x, noise, condition)

    # ------------------------------------ training -------------------------------------------- #
    def _noised(self, x0):
        sigma = at_least_ndim(self.sample_noise_distribution(x0.shape[0]), x0.dim())
        return sigma, torch.randn_like(x0) * sigma * (1. - self.fix_mask)

    def loss(self, x0, condition=None):
        sigma, eps = self._noised(x0)
        cond = self.model["condition"](condition) if condition is not None else None
        err = self.loss_weighting(sigma) * (self.D(x0 + eps, sigma, cond) - x0) ** 2
        return (err * self.loss_weight).mean()

    def update(self, x0, condition=None, **kwargs):
        self.optimizer.zero_grad()
        loss = self.loss(x0, condition)
        loss.backward()
        grad_norm = self._apply_gradients(True, zero_grad=False)       # (this class zeroes BEFORE the backward pass, reference edm.py:101)
        return {"loss": loss.item(), "grad_norm": grad_norm}

    def update_classifier(self, x0, condition):
        sigma, eps = self._noised(x0)
        return self.classifier.update(x0 + eps, at_least_ndim(self.c_noise(sigma).squeeze(), 1), condition)

    # ------------------------------------ sampling -------------------------------------------- #
    def dot_x(self, x, i, use_ema=False, condition_vec_cfg=None, w_cfg: float = 0.0, condition_vec_cg=None,
              w_cg: float = 1.0):
        b = x.shape[0]
        sigma = at_least_ndim(self.sigma_s[i].repeat(b), x.dim())
        unscale = 1. / self.scale_s[i] * (1. - self.fix_mask) + self.fix_mask
        with torch.no_grad():
            if w_cfg != 0.0 and w_cfg != 1.0:
                both = torch.cat([condition_vec_cfg, torch.zeros_like(condition_vec_cfg)], 0)
                rep = [2] + [1] * (x.dim() - 1)
                D = self.D((x * unscale).repeat(*rep), sigma.repeat(*rep), both, use_ema)
                D = w_cfg * D[:b] + (1. - w_cfg) * D[b:]
            else:
                D = self.D(x * unscale, sigma, None if w_cfg == 0.0 else condition_vec_cfg, use_ema)
        log_p = None
        if self.classifier is not None and w_cg != 0.0 and condition_vec_cg is not None:
            noise = at_least_ndim(self.c_noise(sigma).squeeze(), 1)
            log_p, grad = self.classifier.gradients(x * unscale, noise, condition_vec_cg)
            D = D + w_cg * self.scale_s[i] * (sigma ** 2) * grad
        slope = (self.x_weight_s[i] * x - self.D_weight_s[i] * D) * (1. - self.fix_mask)
        return slope, {"log_p": log_p}

    def _native_plan(self, solver, extra_sample_steps):
        return None                                   # only parameterisations with scale_s == 1 have a device plan

    def _run(self, prior, n_samples, sample_steps, extra_sample_steps, use_ema, solver, condition_cfg, mask_cfg, w_cfg,
             condition_cg, w_cg, preserve_history, feed):
        if sample_steps != self.sample_steps:
            self.set_sample_steps(sample_steps)
        N = self.sample_steps
        model = self.model_ema if use_ema else self.model
        cond_cfg = model["condition"](condition_cfg, mask_cfg) if condition_cfg is not None else None
        if prior is None:
            xt = feed.like(torch.empty((n_samples, *self.default_x_shape), device=self.device)) \
                * self.sigma_s[0] * self.scale_s[0]
        else:
            xt = feed.like(prior).to(self.device) * self.sigma_s[0] * self.scale_s[0]
            xt = xt * (1. - self.fix_mask) + prior * self.fix_mask
        history = None
        if preserve_history:
            history = np.empty((n_samples, N + 1, *xt.shape))
            history[:, 0] = xt.cpu().numpy()

        def remask(x):
            return x if prior is None else x * (1. - self.fix_mask) + prior * self.fix_mask

        log = {"log_p": None}
        fused = None
        if not preserve_history and prior is not None:
            plan = self._native_plan(solver, extra_sample_steps)
            if plan is not None:
                from ..engine import dispatch
                fused = dispatch.try_fused_sample(self, model, plan, xt, prior, cond_cfg, w_cfg, w_cg, False, feed)
        if fused is not None:
            xt = fused
        else:
            kw = dict(use_ema=use_ema, condition_vec_cfg=cond_cfg, w_cfg=w_cfg, condition_vec_cg=condition_cg, w_cg=w_cg)
            for i in range(N):
                slope, log = self.dot_x(xt, i, **kw)
                dt = self.t_s[i] - self.t_s[i + 1]
                nxt = remask(xt - slope * dt)
                if solver == "heun" and i != N - 1 and self.sigma_s[i + 1] > 0.005:
                    slope2, log = self.dot_x(nxt, i + 1, **kw)
                    nxt = remask(xt - (slope + slope2) / 2. * dt)
                xt = nxt
                if preserve_history:
                    history[:, i + 1] = xt.cpu().numpy()
            if extra_sample_steps > 0:
                dt = self.t_s[N - 1] - self.t_s[N]
                for _ in range(extra_sample_steps):
                    slope, log = self.dot_x(xt, N - 1, **kw)
                    xt = remask(xt - slope * dt)
        log["sample_history"] = history
        if log["log_p"] is None and self.classifier is not None and condition_cg is not None:
            with torch.no_grad():
                log["log_p"] = self.classifier.logp(xt, at_least_ndim(self.c_noise(self.sigma_s[-1]).squeeze(), 1),
                                                    condition_cg)
        return xt, log

    def sample(self, prior: Optional[torch.Tensor] = None, n_samples: int = 1, sample_steps: int = 5, use_ema: bool = True,
               solver: str = "euler", condition_cfg=None, mask_cfg=None, w_cfg: float = 0.0, condition_cg=None,
               w_cg: float = 0.0, preserve_history: bool = False, **kwargs):
        return self._run(prior, n_samples, sample_steps, 0, use_ema, solver, condition_cfg, mask_cfg, w_cfg, condition_cg,
                         w_cg, preserve_history, _NoiseFeed(kwargs.get("noise", None)))

    def sample_x(self, prior: Optional[torch.Tensor] = None, n_samples: int = 1, sample_steps: int = 5,
                 extra_sample_steps: int = 8, use_ema: bool = True, solver: str = "euler", condition_cfg=None,
                 mask_cfg=None, w_cfg: float = 0.0, condition_cg=None, w_cg: float = 0.0, preserve_history: bool = False,
                 **kwargs):
        return self._run(prior, n_samples, sample_steps, extra_sample_steps, use_ema, solver, condition_cfg, mask_cfg, w_cfg,
                         condition_cg, w_cg, preserve_history, _NoiseFeed(kwargs.get("noise", None)))


class EDM(EDMArchetecture):
    def __init__(self, nn_diffusion: BaseNNDiffusion, nn_condition: Optional[BaseNNCondition] = None,
                 fix_mask=None, loss_weight=None, classifier=None, grad_clip_norm: Optional[float] = None,
                 diffusion_steps: int = 1000, ema_rate: float = 0.995, optim_params: Optional[dict] = None,
                 sigma_data: float = 0.5, sigma_min: float = 0.002, sigma_max: float = 80., rho: float = 7.,
                 P_mean: float = -1.2, P_std: float = 1.2, device: Union[torch.device, str] = "cpu"):
        super().__init__(nn_diffusion, nn_condition, fix_mask, loss_weight, classifier, grad_clip_norm, diffusion_steps,
                         ema_rate, optim_params, device)
        self.sigma_data, self.sigma_min, self.sigma_max, self.rho = sigma_data, sigma_min, sigma_max, rho
        self.P_mean, self.P_std = P_mean, P_std

    def set_sample_steps(self, N: int):
        self.sample_steps = N
        inv = 1 / self.rho
        ramp = torch.arange(N + 1, device=self.device) / N
        self.sigma_s = (self.sigma_max ** inv + ramp * (self.sigma_min ** inv - self.sigma_max ** inv)) ** self.rho
        self.t_s = self.sigma_s
        self.scale_s = torch.ones_like(self.sigma_s)
        self.dot_sigma_s = torch.ones_like(self.sigma_s)
        self.dot_scale_s = torch.zeros_like(self.sigma_s)
        self.x_weight_s = self.dot_sigma_s / self.sigma_s + self.dot_scale_s / self.scale_s
        self.D_weight_s = self.dot_sigma_s / self.sigma_s * self.scale_s

    def c_skip(self, sigma):
        return self.sigma_data ** 2 / (self.sigma_data ** 2 + sigma ** 2)

    def c_out(self, sigma):
        return sigma * self.sigma_data / (self.sigma_data ** 2 + sigma ** 2).sqrt()

    def c_in(self, sigma):
        return 1 / (self.sigma_data ** 2 + sigma ** 2).sqrt()

    def c_noise(self, sigma):
        return 0.25 * sigma.log()

    def loss_weighting(self, sigma):
        return (self.sigma_data ** 2 + sigma ** 2) / ((sigma * self.sigma_data) ** 2)

    def sample_noise_distribution(self, N):
        return (torch.randn(N, device=self.device) * self.P_std + self.P_mean).exp()

    def sample_scale_distribution(self, N):
        return torch.ones(N, device=self.device)

    def _native_plan(self, solver, extra_sample_steps):
        if type(self) is not EDM:
            return None
        from ..engine.plan import build_legacy_edm_plan
        key = ("legacy_edm", solver, extra_sample_steps, self.sample_steps, self.sigma_data, self.sigma_min, self.sigma_max, self.rho)
        return self._cached_plan(key, lambda: build_legacy_edm_plan(self.sigma_data, self.sigma_s, solver, extra_sample_steps))
