"""Legacy ``DDPM`` solver (imported by the dp_* / dbc_* pipelines as ``cleandiffuser.diffusion.ddpm.DDPM``).

Contract: reference diffusion/ddpm.py:17-350 -- beta schedule ("cosine" | "linear") -> alpha, bar_alpha; ancestral sampling
over ALL ``diffusion_steps`` (``sample_steps`` is overridden with a warning); ``predict_function`` applies
CFG, classifier guidance, clipping via bar_alpha and the fix-mask on the prediction; ``sample_x`` appends
``extra_sample_steps`` noise-free updates at t = 0 (Diffusion-X).  The returned ``log`` is whatever the last
``predict_function`` call produced (``{"log_p": ...}``), as in the reference.

Execution: PyTorch executor; shares the backbone dispatch, so a fused backbone forward is used per step when available.
"""
import warnings
from typing import Optional, Union

import numpy as np
import torch
import torch.nn as nn

from ..nn_condition import BaseNNCondition
from ..nn_diffusion import BaseNNDiffusion
from ..utils import at_least_ndim, cosine_beta_schedule, linear_beta_schedule
from .basic import DiffusionModel
from .diffusionsde import _NoiseFeed


class DDPM(DiffusionModel):
    def __init__(self, nn_diffusion: BaseNNDiffusion, nn_condition: Optional[BaseNNCondition] = None,
                 fix_mask=None, loss_weight=None, classifier=None, grad_clip_norm: Optional[float] = None,
                 diffusion_steps: int = 1000, ema_rate: float = 0.995, optim_params: Optional[dict] = None,
                 x_max: Optional[torch.Tensor] = None, x_min: Optional[torch.Tensor] = None,
                 predict_noise: bool = True, beta_schedule: str = "cosine",
                 beta_schedule_params: Optional[dict] = None, device: Union[torch.device, str] = "cpu"):
        super().__init__(nn_diffusion, nn_condition, fix_mask, loss_weight, classifier, grad_clip_norm,
                         diffusion_steps, ema_rate, optim_params, device)
        self.predict_noise = predict_noise
        params = dict(beta_schedule_params or {})
        params["T"] = self.diffusion_steps
        if beta_schedule == "linear":
            beta = linear_beta_schedule(**params)
        elif beta_schedule == "cosine":
            beta = cosine_beta_schedule(**params)
        else:
            raise ValueError(f"Unknown beta schedule: {beta_schedule}")
        self.beta = torch.tensor(beta, device=self.device, dtype=torch.float32)
        self.alpha = 1 - self.beta
        self.bar_alpha = torch.cumprod(self.alpha.clone(), 0)
        self.x_max, self.x_min = x_max, x_min

    @property
    def clip_pred(self):
        return (self.x_max is not None) or (self.x_min is not None)

    # ------------------------------------ training -------------------------------------------- #
    def add_noise(self, x0, t=None, eps=None):
        t = torch.randint(self.diffusion_steps, (x0.shape[0],), device=self.device) if t is None else t
        eps = torch.randn_like(x0) if eps is None else eps
        bar = at_least_ndim(self.bar_alpha[t], x0.dim())
        xt = x0 * bar.sqrt() + eps * (1 - bar).sqrt()
        return xt * (1. - self.fix_mask) + x0 * self.fix_mask, t, eps

    def loss(self, x0, condition=None):
        xt, t, eps = self.add_noise(x0)
        cond = self.model["condition"](condition) if condition is not None else None
        target = eps if self.predict_noise else x0
        err = (self.model["diffusion"](xt, t, cond) - target) ** 2
        return (err * self.loss_weight * (1 - self.fix_mask)).mean()

    def update(self, x0, condition=None, update_ema=True, **kwargs):
        loss = self._loss_backward(x0, condition)
        grad_norm = self._apply_gradients(update_ema)
        return {"loss": loss.item(), "grad_norm": grad_norm}

    def update_classifier(self, x0, condition):
        xt, t, _ = self.add_noise(x0)
        return self.classifier.update(xt, t, condition)

    # ------------------------------------ sampling -------------------------------------------- #
    def predict_function(self, x, t, bar_alpha, use_ema=False, requires_grad=False, condition_vec_cfg=None,
                         w_cfg: float = 0.0, condition_vec_cg=None, w_cg: float = 1.0):
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

        log_p = None
        if self.classifier is not None and w_cg != 0.0 and condition_vec_cg is not None:
            log_p, grad = self.classifier.gradients(x.clone(), t, condition_vec_cg)
            if self.predict_noise:
                pred = pred - w_cg * (1 - bar_alpha).sqrt() * grad
            else:
                pred = pred + w_cg * (1 - bar_alpha) / bar_alpha.sqrt() * grad

        if self.predict_noise:
            if self.clip_pred:
                hi = (x - bar_alpha.sqrt() * self.x_min) / (1 - bar_alpha).sqrt() if self.x_min is not None else None
                lo = (x - bar_alpha.sqrt() * self.x_max) / (1 - bar_alpha).sqrt() if self.x_max is not None else None
                pred = pred.clip(lo, hi)
            pred = pred * (1 - self.fix_mask)
        else:
            if self.clip_pred:
                pred = pred.clip(self.x_min, self.x_max)
            pred = pred * (1 - self.fix_mask) + x * self.fix_mask
        return pred, {"log_p": log_p}

    def _posterior_step(self, xt, pred, t):
        """Ancestral update mean at integer step t (reference ddpm.py:230-238)."""
        bar, alpha, beta = self.bar_alpha[t], self.alpha[t], self.beta[t]
        bar_prev = self.bar_alpha[t - 1] if t > 0 else torch.tensor(1.0, device=self.device)
        if self.predict_noise:
            mean = 1 / alpha.sqrt() * (xt - beta / (1 - bar).sqrt() * pred)
        else:
            mean = 1 / (1 - bar) * (alpha.sqrt() * (1 - bar_prev) * xt + beta * bar_prev.sqrt() * pred)
        return mean, (beta * (1 - bar_prev) / (1 - bar)).sqrt()

    def _run(self, prior, n_samples, sample_steps, extra_sample_steps, use_ema, temperature, condition_cfg, mask_cfg,
             w_cfg, condition_cg, w_cg, requires_grad, preserve_history, feed):
        log = {"sample_history": np.empty((n_samples, sample_steps + 1, *prior.shape)) if preserve_history else None}
        model = self.model_ema if use_ema else self.model
        if sample_steps != self.diffusion_steps:
            warnings.warn("sample_steps != diffusion_steps, sample_steps will be set to diffusion_steps.")
        xt = feed.like(prior).to(self.device) * temperature
        xt = xt * (1. - self.fix_mask) + prior * self.fix_mask
        with torch.set_grad_enabled(requires_grad):
            cond_cfg = model["condition"](condition_cfg, mask_cfg) if condition_cfg is not None else None
        if not preserve_history:                                   # whole loop in one launch when the backbone compiles
            from ..engine import dispatch, plan as _plan
            plan = self._cached_plan(("legacy_ddpm", bool(self.predict_noise), extra_sample_steps),
                                     lambda: _plan.build_legacy_ddpm_plan(self.beta, self.alpha, self.bar_alpha,
                                                                          self.predict_noise, extra_sample_steps))
            fused = dispatch.try_fused_legacy_ddpm(self, model, plan, xt, prior, cond_cfg, w_cfg, w_cg, requires_grad, feed)
            if fused is not None:
                log = {"log_p": None}
                if self.classifier is not None and condition_cg is not None:
                    with torch.no_grad():
                        t0 = torch.zeros((n_samples,), dtype=torch.long, device=self.device)
                        log["log_p"] = self.classifier.logp(fused, t0, condition_cg)
                return fused, log
        kw = dict(use_ema=use_ema, requires_grad=requires_grad, condition_vec_cfg=cond_cfg,
                  condition_vec_cg=condition_cg, w_cfg=w_cfg, w_cg=w_cg)
        feed.reserve(xt, self.diffusion_steps - 1)     # (as the whole-loop executor draws them: _NoiseFeed.reserve)
        for t in range(self.diffusion_steps - 1, -1, -1):
            t_batch = torch.tensor(t, device=self.device, dtype=torch.long).repeat(n_samples)
            pred, log = self.predict_function(xt, t_batch, self.bar_alpha[t], **kw)
            xt, std = self._posterior_step(xt, pred, t)
            if t != 0:
                xt = xt + std * feed.like(xt)
            xt = xt * (1. - self.fix_mask) + prior * self.fix_mask
        if extra_sample_steps > 0:
            t_batch = torch.tensor(0, device=self.device, dtype=torch.long).repeat(n_samples)
            for _ in range(extra_sample_steps):
                pred, log = self.predict_function(xt, t_batch, self.bar_alpha[0], **kw)
                xt, _ = self._posterior_step(xt, pred, 0)
                xt = xt * (1. - self.fix_mask) + prior * self.fix_mask
        if log["log_p"] is None and self.classifier is not None and condition_cg is not None:
            with torch.no_grad():
                t0 = torch.zeros((n_samples,), dtype=torch.long, device=self.device)
                log["log_p"] = self.classifier.logp(xt, t0, condition_cg)
        return xt, log

    def sample(self, prior: Optional[torch.Tensor] = None, n_samples: int = 1, sample_steps: int = None,
               use_ema: bool = True, temperature: float = 1.0, condition_cfg=None, mask_cfg=None, w_cfg: float = 0.0,
               condition_cg=None, w_cg: float = 0.0, requires_grad: bool = False, preserve_history: bool = False,
               **kwargs):
        return self._run(prior, n_samples, sample_steps, 0, use_ema, temperature, condition_cfg, mask_cfg, w_cfg,
                         condition_cg, w_cg, requires_grad, preserve_history, _NoiseFeed(kwargs.get("noise", None)))

    def sample_x(self, prior: Optional[torch.Tensor] = None, n_samples: int = 1, sample_steps: int = None,
                 extra_sample_steps: int = 8, use_ema: bool = True, temperature: float = 1.0, condition_cfg=None,
                 mask_cfg=None, w_cfg: float = 0.0, condition_cg=None, w_cg: float = 0.0, requires_grad: bool = False,
                 preserve_history: bool = False, **kwargs):
        return self._run(prior, n_samples, sample_steps, extra_sample_steps, use_ema, temperature, condition_cfg,
                         mask_cfg, w_cfg, condition_cg, w_cg, requires_grad, preserve_history,
                         _NoiseFeed(kwargs.get("noise", None)))
