"""Legacy ``DPMSolver`` (imported by the dp_* / dbc_* pipelines as ``cleandiffuser.diffusion.dpmsolver.DPMSolver``).

Contract: reference diffusion/dpmsolver.py:13-49 (sampler table), :66-89 (one-step estimates), :92-203 (ctor, VP schedule
alpha(t) for "linear" | "cosine", training), :206-268 (``predict_function``: CFG, eps<->x0 conversion to what the sampler
consumes, classifier guidance, clipping, fix-mask applied to the *prediction*), :435-528 (``sample``) and :531-623
(``sample_x`` = Diffusion-X tail for first-order samplers).  Time grid: ``t_i = ((S-i)/S t1^(1/k) + i/S t0^(1/k))^k``.

Execution: on a ROCm device the whole loop is one native call (``engine/plan.py:build_legacy_dpmsolver_plan`` -> linear step
records with the MASK_PRED flag); otherwise the PyTorch loop below, which evaluates the reference's expressions verbatim.
"""
from typing import Optional, Union

import numpy as np
import torch
import torch.nn as nn

from ..engine.plan import LEGACY_DPM_SAMPLERS
from ..nn_condition import BaseNNCondition
from ..nn_diffusion import BaseNNDiffusion
from ..utils import at_least_ndim
from .basic import DiffusionModel
from .diffusionsde import _NoiseFeed

SAMPLER_CONFIG = {name: {"predict_noise": eps, "order": order} for name, (eps, order) in LEGACY_DPM_SAMPLERS.items()}


def _one_step(xt, pred, i, alphas, sigmas, h, sampler, feed):
    family = "ode_dpm" if sampler == "ddim" else sampler[:-2]
    if family == "ode_dpm":
        return alphas[i] / alphas[i - 1] * xt - sigmas[i] * torch.expm1(h[i]) * pred
    if family == "sde_dpm":
        return (alphas[i] / alphas[i - 1] * xt - 2. * sigmas[i] * torch.expm1(h[i]) * pred +
                sigmas[i] * torch.expm1(2. * h[i]).sqrt() * feed.like(xt))
    if family == "ode_dpmpp":
        return sigmas[i] / sigmas[i - 1] * xt - alphas[i] * torch.expm1(-h[i]) * pred
    if family == "sde_dpmpp":
        return (sigmas[i] / sigmas[i - 1] * (-h[i]).exp() * xt - alphas[i] * torch.expm1(-2. * h[i]) * pred +
                sigmas[i] * (-1. * torch.expm1(-2. * h[i])).sqrt() * feed.like(xt))
    raise ValueError(f"Unknown sampler: {sampler}.")


class DPMSolver(DiffusionModel):
    def __init__(self, nn_diffusion: BaseNNDiffusion, nn_condition: Optional[BaseNNCondition] = None,
                 fix_mask=None, loss_weight=None, classifier=None, grad_clip_norm: Optional[float] = None,
                 diffusion_steps: int = 1000, ema_rate: float = 0.995, optim_params: Optional[dict] = None,
                 x_max: Optional[torch.Tensor] = None, x_min: Optional[torch.Tensor] = None, predict_noise: bool = False,
                 noise_schedule: str = "linear", t_eps: float = 1e-3, device: Union[torch.device, str] = "cpu"):
        super().__init__(nn_diffusion, nn_condition, fix_mask, loss_weight, classifier, grad_clip_norm, diffusion_steps,
                         ema_rate, optim_params, device)
        self.predict_noise, self.noise_schedule, self.t_eps = predict_noise, noise_schedule, t_eps
        self.x_max, self.x_min = x_max, x_min

    @property
    def clip_pred(self):
        return (self.x_max is not None) or (self.x_min is not None)

    @property
    def supported_samplers(self):
        return list(SAMPLER_CONFIG.keys())

    @property
    def t_range(self):
        if self.noise_schedule == "linear":
            return self.t_eps, 1.
        if self.noise_schedule == "cosine":
            return self.t_eps, 0.9946
        raise ValueError(f"noise_schedule should be 'linear' or 'cosine', but got {self.noise_schedule}.")

    def alpha_schedule(self, t):
        if self.noise_schedule == "linear":
            beta0, beta1 = 0.1, 20
            return (-(beta1 - beta0) / 4 * (t ** 2) - beta0 / 2 * t).exp()
        if self.noise_schedule == "cosine":
            s = 0.008
            return ((torch.cos(np.pi / 2 * (t + s) / (1 + s))).log() - np.log(np.cos(np.pi / 2 * s / (1 + s)))).exp()
        raise ValueError(f"noise_schedule should be 'linear' or 'cosine', but got {self.noise_schedule}.")

    # ------------------------------------ training -------------------------------------------- #
    def add_noise(self, x0, t=None, eps=None):
        if t is None:
            t = torch.rand((x0.shape[0],), device=self.device)
            t = self.t_range[0] + t * (self.t_range[1] - self.t_range[0])
        eps = torch.randn_like(x0) if eps is None else eps
        alpha = self.alpha_schedule(at_least_ndim(t, x0.dim()))
        sigma = (1 - alpha ** 2).sqrt()
        xt = x0 * alpha + sigma * eps
        return xt * (1. - self.fix_mask) + x0 * self.fix_mask, t, eps

    def loss(self, x0, condition=None):
        xt, t, eps = self.add_noise(x0)
        cond = self.model["condition"](condition) if condition is not None else None
        target = eps if self.predict_noise else x0
        err = (self.model["diffusion"](xt, t, cond) - target) ** 2
        return (err * self.loss_weight * (1 - self.fix_mask)).mean()

    def update(self, x0, condition=None, update_ema=True, **kwargs):
        loss = self.loss(x0, condition)
        loss.backward()
        grad_norm = self._apply_gradients(update_ema)
        return {"loss": loss.item(), "grad_norm": grad_norm}

    def update_classifier(self, x0, condition):
        xt, t, _ = self.add_noise(x0)
        return self.classifier.update(xt, t, condition)

    # ------------------------------------ sampling -------------------------------------------- #
    def predict_function(self, x, t, alpha, sigma, use_ema=False, requires_grad=False, predict_noise=False,
                         condition_vec_cfg=None, w_cfg: float = 0.0, condition_vec_cg=None, w_cg: float = 1.0):
        b = x.shape[0]
        model = self.model_ema if use_ema else self.model
        with torch.set_grad_enabled(requires_grad):
            if w_cfg != 0.0 and w_cfg != 1.0:
                both = torch.cat([condition_vec_cfg, torch.zeros_like(condition_vec_cfg)], 0)
                out = model["diffusion"](x.repeat(2, *([1] * (x.dim() - 1))), t.repeat(2), both)
                pred = w_cfg * out[:b] + (1. - w_cfg) * out[b:]
            elif w_cfg == 0.0:
                pred = model["diffusion"](x, t, None)
            else:
                pred = model["diffusion"](x, t, condition_vec_cfg)
        if self.predict_noise and not predict_noise:
            pred = (x - sigma * pred) / alpha
        elif not self.predict_noise and predict_noise:
            pred = (x - alpha * pred) / sigma

        log_p = None
        if self.classifier is not None and w_cg != 0.0 and condition_vec_cg is not None:
            log_p, grad = self.classifier.gradients(x.clone(), t, condition_vec_cg)
            pred = pred - w_cg * sigma * grad if predict_noise else pred + w_cg * ((sigma ** 2) / alpha) * grad

        if predict_noise:
            if self.clip_pred:
                hi = (x - alpha * self.x_min) / sigma if self.x_min is not None else None
                lo = (x - alpha * self.x_max) / sigma if self.x_max is not None else None
                pred = pred.clip(lo, hi)
            pred = pred * (1 - self.fix_mask)
        else:
            if self.clip_pred:
                pred = pred.clip(self.x_min, self.x_max)
            pred = pred * (1 - self.fix_mask) + x * self.fix_mask
        return pred, {"log_p": log_p}

    def _run(self, prior, n_samples, sample_steps, extra_sample_steps, use_ema, temperature, kappa, sampler, condition_cfg,
             mask_cfg, w_cfg, condition_cg, w_cg, requires_grad, preserve_history, feed):
        assert sampler in self.supported_samplers, f"Sampler '{sampler}' is not supported."
        cfg = SAMPLER_CONFIG[sampler]
        log = {"sample_history": np.empty((n_samples, sample_steps + 1, *prior.shape)) if preserve_history else None}
        model = self.model_ema if use_ema else self.model
        idx = torch.arange(sample_steps + 1)
        t = (((sample_steps - idx) / sample_steps * self.t_range[1] ** (1 / kappa) +
              idx / sample_steps * self.t_range[0] ** (1 / kappa)) ** kappa).to(self.device)
        alphas = self.alpha_schedule(t)
        sigmas = (1 - alphas ** 2).sqrt()
        log_snr = (alphas / sigmas).log()
        h = torch.zeros_like(log_snr)
        h[1:] = log_snr[1:] - log_snr[:-1]

        xt = feed.like(prior).to(self.device) * temperature
        xt = xt * (1. - self.fix_mask) + prior * self.fix_mask
        if preserve_history:
            log["sample_history"][:, 0] = xt.cpu().numpy()
        with torch.set_grad_enabled(requires_grad):
            cond_cfg = model["condition"](condition_cfg, mask_cfg) if condition_cfg is not None else None

        fused = None
        if not preserve_history:
            from ..engine import dispatch, plan as _plan
            plan = self._cached_plan((sampler, sample_steps, extra_sample_steps, float(kappa)), lambda: _plan.build_legacy_dpmsolver_plan(
                t, alphas, sigmas, sampler, sample_steps, extra_sample_steps))
            fused = dispatch.try_fused_sample(self, model, plan, xt, prior, cond_cfg, w_cfg, w_cg, requires_grad, feed)
        if fused is not None:
            xt, log = fused, {"sample_history": None, "log_p": None}
        else:
            kw = dict(use_ema=use_ema, requires_grad=requires_grad, predict_noise=cfg["predict_noise"],
                      condition_vec_cfg=cond_cfg, condition_vec_cg=condition_cg, w_cfg=w_cfg, w_cg=w_cg)
            if not preserve_history:
                feed.reserve(xt, plan.n_noise)         # (as the whole-loop executor draws them: _NoiseFeed.reserve)
            buffer = []
            for i in range(1, sample_steps + 1):
                pred, log_i = self.predict_function(xt, t[i - 1].repeat(n_samples), alphas[i - 1], sigmas[i - 1], **kw)
                log.update(log_i)
                if cfg["order"] == 2:
                    buffer.append(pred.clone())
                if cfg["order"] == 2 and i > 1:
                    r = h[i - 1] / h[i]
                    pred = (1 + 0.5 / r) * buffer[-1] - 0.5 / r * buffer[-2]
                xt = _one_step(xt, pred, i, alphas, sigmas, h, sampler, feed)
                xt = xt * (1. - self.fix_mask) + prior * self.fix_mask
                if preserve_history:
                    log["sample_history"][:, 1] = xt.cpu().numpy()         # reference quirk: always slot 1
            if cfg["order"] == 1:
                s = sample_steps
                for _ in range(extra_sample_steps):
                    pred, log_i = self.predict_function(xt, t[s - 1].repeat(n_samples), alphas[s - 1], sigmas[s - 1], **kw)
                    log.update(log_i)
                    xt = _one_step(xt, pred, s, alphas, sigmas, h, sampler, feed)
                    xt = xt * (1. - self.fix_mask) + prior * self.fix_mask
        if log.get("log_p") is None and self.classifier is not None and condition_cg is not None:
            with torch.no_grad():
                log["log_p"] = self.classifier.logp(xt, t[-1].repeat(n_samples), condition_cg)
        if self.clip_pred:
            xt = xt.clip(self.x_min, self.x_max)
        return xt, log

    def sample(self, prior: Optional[torch.Tensor] = None, n_samples: int = 1, sample_steps: int = 5, use_ema: bool = True,
               temperature: float = 1.0, kappa: float = 1.0, sampler: str = "ddim", condition_cfg=None, mask_cfg=None,
               w_cfg: float = 0.0, condition_cg=None, w_cg: float = 0.0, requires_grad: bool = False,
               preserve_history: bool = False, **kwargs):
        return self._run(prior, n_samples, sample_steps, 0, use_ema, temperature, kappa, sampler, condition_cfg, mask_cfg,
                         w_cfg, condition_cg, w_cg, requires_grad, preserve_history, _NoiseFeed(kwargs.get("noise", None)))

    def sample_x(self, prior: Optional[torch.Tensor] = None, n_samples: int = 1, sample_steps: int = 5,
                 extra_sample_steps: int = 8, use_ema: bool = True, temperature: float = 1.0, kappa: float = 1.0,
                 sampler: str = "ddim", condition_cfg=None, mask_cfg=None, w_cfg: float = 0.0, condition_cg=None,
                 w_cg: float = 0.0, requires_grad: bool = False, preserve_history: bool = False, **kwargs):
        return self._run(prior, n_samples, sample_steps, extra_sample_steps, use_ema, temperature, kappa, sampler,
                         condition_cfg, mask_cfg, w_cfg, condition_cg, w_cg, requires_grad, preserve_history,
                         _NoiseFeed(kwargs.get("noise", None)))
