"""ContinuousEDM -- Karras et al. EDM with sigma(t) = t, scale 1, Euler / Heun ODE samplers.

Contract: reference diffusion/newedm.py (ctor :69-103, preconditioning :130-148, loss :165-173, guidance :217-284,
``sample`` :286-438).  Reference behaviours kept: initial noise ``randn * sigma_max * temperature`` (:376), rho-schedule
``(smin^(1/rho) + k/S (sfwd^(1/rho) - smin^(1/rho)))^rho`` (:386-388), the network sees ``c_in * x`` and ``c_noise = ln(sigma)/4``,
Heun's second evaluation at ``t * sigma_{i-1} / sigma_i`` and none on the last step (:413-421), classifier guidance only when a
condition is given (:229), final ``log_p`` at ``sigma_min`` whenever a classifier exists (:430-434).

Execution: PyTorch executor (every backbone).  The fused gfx950 loop does not cover EDM yet.
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


class ContinuousEDM(DiffusionModel):
    def __init__(self, nn_diffusion: BaseNNDiffusion, nn_condition: Optional[BaseNNCondition] = None,
                 fix_mask=None, loss_weight=None, classifier=None, grad_clip_norm: Optional[float] = None,
                 ema_rate: float = 0.995, optim_params: Optional[dict] = None, sigma_data: float = 0.5,
                 sigma_min: float = 0.002, sigma_max: float = 80., rho: float = 7., P_mean: float = -1.2,
                 P_std: float = 1.2, x_max: Optional[torch.Tensor] = None, x_min: Optional[torch.Tensor] = None,
                 device: Union[torch.device, str] = "cpu"):
        super().__init__(nn_diffusion, nn_condition, fix_mask, loss_weight, classifier, grad_clip_norm, 0, ema_rate,
                         optim_params, device)
        self.sigma_data, self.sigma_min, self.sigma_max = sigma_data, sigma_min, sigma_max
        self.rho, self.P_mean, self.P_std = rho, P_mean, P_std
        self.x_max = x_max.to(device) if isinstance(x_max, torch.Tensor) else x_max
        self.x_min = x_min.to(device) if isinstance(x_min, torch.Tensor) else x_min
        self.t_diffusion = [sigma_min, sigma_max]

    @property
    def supported_solvers(self):
        return ["euler", "heun"]

    @property
    def clip_pred(self):
        return (self.x_max is not None) or (self.x_min is not None)

    # ------------------------------- preconditioning ------------------------------------------ #
    def c_skip(self, sigma):
        return self.sigma_data ** 2 / (self.sigma_data ** 2 + sigma ** 2)

    def c_out(self, sigma):
        return sigma * self.sigma_data / (self.sigma_data ** 2 + sigma ** 2).sqrt()

    def c_in(self, sigma):
        return 1 / (self.sigma_data ** 2 + sigma ** 2).sqrt()

    def c_noise(self, sigma):
        return 0.25 * sigma.log()

    def D(self, x, sigma, condition=None, model=None):
        """Denoiser D(x; sigma) = c_skip x + c_out F(c_in x, c_noise, cond)."""
        model = self.model if model is None else model
        skip, out, inn = (at_least_ndim(f(sigma), x.dim()) for f in (self.c_skip, self.c_out, self.c_in))
        return skip * x + out * model["diffusion"](inn * x, self.c_noise(sigma), condition)

    # ------------------------------------ training -------------------------------------------- #
    def add_noise(self, x0, t=None, eps=None):
        t = (torch.randn((x0.shape[0],), device=self.device) * self.P_std + self.P_mean).exp() if t is None else t
        eps = torch.randn_like(x0) if eps is None else eps
        xt = 1. * x0 + at_least_ndim(t, x0.dim()) * eps
        return (1. - self.fix_mask) * xt + self.fix_mask * x0, t, eps

    def loss(self, x0, condition=None):
        xt, t, _ = self.add_noise(x0)
        cond = self.model["condition"](condition) if condition is not None else None
        err = (self.D(xt, t, cond) - x0) ** 2
        w = at_least_ndim((t ** 2 + self.sigma_data ** 2) / ((t * self.sigma_data) ** 2), x0.dim())
        return (err * self.loss_weight * (1 - self.fix_mask) * w).mean()

    def update(self, x0, condition=None, update_ema=True, **kwargs):
        loss = self._loss_backward(x0, condition)
        grad_norm = self._apply_gradients(update_ema)
        return {"loss": loss.item(), "grad_norm": grad_norm}

    def update_classifier(self, x0, condition):
        xt, t, _ = self.add_noise(x0)
        return self.classifier.update(xt, t.log() / 4., condition)

    # ------------------------------------ guidance -------------------------------------------- #
    def classifier_guidance(self, xt, t, sigma, model, condition=None, w: float = 1.0, pred=None):
        if pred is None:
            pred = self.D(xt, t, None, model)
        if self.classifier is None or w == 0.0 or condition is None:
            return pred, None
        log_p, grad = self.classifier.gradients(xt.clone(), t.log() / 4., condition)
        return pred + w * (at_least_ndim(sigma, pred.dim()) ** 2) * grad, log_p

    def classifier_free_guidance(self, xt, t, model, condition=None, w: float = 1.0, pred=None, pred_uncond=None,
                                 requires_grad: bool = False):
        with torch.set_grad_enabled(requires_grad):
            if w != 0.0 and w != 1.0:
                if pred is None or pred_uncond is None:
                    b = xt.shape[0]
                    both = torch.cat([condition, torch.zeros_like(condition)], 0)
                    out = self.D(xt.repeat(2, *([1] * (xt.dim() - 1))), t.repeat(2), both, model)
                    pred, pred_uncond = out[:b], out[b:]
            elif w == 0.0:
                pred, pred_uncond = 0., self.D(xt, t, None, model)
            else:
                pred, pred_uncond = self.D(xt, t, condition, model), 0.
        return w * pred + (1 - w) * pred_uncond

    def guided_sampling(self, xt, t, sigma, model, condition_cfg=None, w_cfg: float = 0.0, condition_cg=None,
                        w_cg: float = 0.0, requires_grad: bool = False):
        pred = self.classifier_free_guidance(xt, t, model, condition_cfg, w_cfg, None, None, requires_grad)
        return self.classifier_guidance(xt, t, sigma, model, condition_cg, w_cg, pred)

    # ------------------------------------ sampling -------------------------------------------- #
    def sample(self, prior: torch.Tensor, solver: str = "euler", n_samples: int = 1, sample_steps: int = 5,
               use_ema: bool = True, temperature: float = 1.0, condition_cfg=None, mask_cfg=None, w_cfg: float = 0.0,
               condition_cg=None, w_cg: float = 0.0, diffusion_x_sampling_steps: int = 0,
               warm_start_reference: Optional[torch.Tensor] = None, warm_start_forward_level: float = 0.3,
               requires_grad: bool = False, preserve_history: bool = False, **kwargs):
        assert solver in ["euler", "heun"], f"Solver {solver} is not supported. Use 'euler' or 'heun' instead."
        feed = _NoiseFeed(kwargs.get("noise", None))
        log = {"sample_history": np.empty((n_samples, sample_steps + 1, *prior.shape)) if preserve_history else None}
        model = self.model_ema if use_ema else self.model
        prior = prior.to(self.device)

        if isinstance(warm_start_reference, torch.Tensor) and warm_start_forward_level > 0.:
            top_sigma = self.sigma_min + (self.sigma_max - self.sigma_min) * warm_start_forward_level
            xt = warm_start_reference + top_sigma * feed.like(warm_start_reference)
        else:
            top_sigma = self.sigma_max
            xt = feed.like(prior) * self.sigma_max * temperature
        xt = xt * (1. - self.fix_mask) + prior * self.fix_mask
        if preserve_history:
            log["sample_history"][:, 0] = xt.cpu().numpy()

        with torch.set_grad_enabled(requires_grad):
            cond_cfg = model["condition"](condition_cfg, mask_cfg) if condition_cfg is not None else None

        inv_rho = 1 / self.rho
        ramp = torch.arange(sample_steps + 1, device=self.device) / sample_steps
        sigmas = (self.sigma_min ** inv_rho + ramp * (top_sigma ** inv_rho - self.sigma_min ** inv_rho)) ** self.rho

        if not preserve_history:
            from ..engine import dispatch
            from ..engine.plan import build_edm_plan
            plan = self._cached_plan((solver, sample_steps, diffusion_x_sampling_steps, float(top_sigma)), lambda: build_edm_plan(
                self.sigma_data, self.sigma_min, top_sigma, self.rho, sample_steps, solver, diffusion_x_sampling_steps))
            fused = dispatch.try_fused_edm(self, model, plan, xt, prior, cond_cfg, w_cfg, w_cg, requires_grad, feed,
                                           condition_cg=condition_cg)
            if fused is not None:
                return self._finish_sample(fused, n_samples, condition_cg, log)

        def denoise(x, t, sigma):
            pred, _ = self.guided_sampling(x, t, sigma, model, cond_cfg, w_cfg, condition_cg, w_cg, requires_grad)
            return pred.clip(self.x_min, self.x_max) if self.clip_pred else pred

        def remask(x):
            return x * (1. - self.fix_mask) + prior * self.fix_mask

        for i in reversed([1] * diffusion_x_sampling_steps + list(range(1, sample_steps + 1))):
            t = torch.full((n_samples,), sigmas[i], dtype=torch.float32, device=self.device)
            slope = (xt - denoise(xt, t, sigmas[i])) / at_least_ndim(sigmas[i], xt.dim())
            dt = sigmas[i] - sigmas[i - 1]
            nxt = remask(xt - slope * dt)
            if solver == "heun" and i > 1:
                slope2 = (nxt - denoise(nxt, t / sigmas[i] * sigmas[i - 1], sigmas[i - 1])) \
                    / at_least_ndim(sigmas[i - 1], xt.dim())
                nxt = remask(xt - (slope + slope2) / 2. * dt)
            xt = nxt
            if preserve_history:
                log["sample_history"][:, sample_steps - i + 1] = xt.cpu().numpy()

        return self._finish_sample(xt, n_samples, condition_cg, log)

    def _finish_sample(self, xt, n_samples, condition_cg, log):
        if self.classifier is not None:
            with torch.no_grad():
                t = torch.ones((n_samples,), dtype=torch.long, device=self.device) * self.sigma_min
                log["log_p"] = self.classifier.logp(xt, t.log() / 4., condition_cg)
        if self.clip_pred:
            xt = xt.clip(self.x_min, self.x_max)
        return xt, log
