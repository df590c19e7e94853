"""Rectified-flow solvers: straight-line transport between a source sample x1 (Gaussian unless given) and data x0.

Contract: reference diffusion/rectifiedflow.py:16-337 (``DiscreteRectifiedFlow``: time grid ``t_diffusion`` from the
discretization registry with eps = 0, velocity target ``x0 - x1``, integer timesteps fed to the backbone) and :340-632
(``ContinuousRectifiedFlow``: t ~ U(0,1), float timesteps).  Sampling is explicit Euler on dx/dt = -v from t = 1 (or the
warm-start level) to 0: ``x <- x + (t_i - t_{i-1}) v(x, t_i)``, CFG on v, fix-mask after every step, one final clip.
No classifier guidance.

Execution: on a ROCm device the whole loop is one native call -- each Euler step is a ``linear`` record (x <- 1 x - (-dt) V,
V = the raw prediction, no per-step clipping; ``engine/plan.py:build_flow_plan``); otherwise the PyTorch loop below.
"""
from typing import Callable, Optional, Union

import numpy as np
import torch
import torch.nn as nn

from ..nn_condition import BaseNNCondition
from ..nn_diffusion import BaseNNDiffusion
from ..utils import SUPPORTED_DISCRETIZATIONS, SUPPORTED_SAMPLING_STEP_SCHEDULE, at_least_ndim
from .basic import DiffusionModel
from .diffusionsde import _NoiseFeed


class _RectifiedFlow(DiffusionModel):
    """Shared machinery; subclasses decide what a timestep is (grid index vs. float) and how training times are drawn."""

    _integer_time = False

    @property
    def supported_solvers(self):
        return ["euler"]

    @property
    def clip_pred(self):
        return (self.x_max is not None) or (self.x_min is not None)

    def _training_times(self, n):          # -> (value fed to the backbone, interpolation weight)
        raise NotImplementedError

    def loss(self, x0, x1=None, condition=None):
        if x1 is None:
            x1 = torch.randn_like(x0)
        else:
            assert x0.shape == x1.shape, "x0 and x1 must have the same shape"
        t, w = self._training_times(x0.shape[0])
        w = at_least_ndim(w, x0.dim())
        xt = self._interpolate(x0, x1, w)
        xt = xt * (1. - self.fix_mask) + x0 * self.fix_mask
        cond = self.model["condition"](condition) if condition is not None else None
        err = (self.model["diffusion"](xt, t, cond) - (x0 - x1)) ** 2
        return (err * self.loss_weight * (1 - self.fix_mask)).mean()

    def update(self, x0, condition=None, update_ema=True, x1=None, **kwargs):
        loss = self.loss(x0, x1, condition)
        loss.backward()
        grad_norm = self._apply_gradients(update_ema)
        return {"loss": loss.item(), "grad_norm": grad_norm}

    def _velocity(self, model, xt, t, cond, w_cfg):
        if w_cfg != 0.0 and w_cfg != 1.0 and cond is not None:
            both = torch.cat([cond, torch.zeros_like(cond)], 0)
            v_c, v_u = model["diffusion"](xt.repeat(2, *([1] * (xt.dim() - 1))), t.repeat(2), both).chunk(2, dim=0)
            return w_cfg * v_c + (1 - w_cfg) * v_u
        return model["diffusion"](xt, t, None if (w_cfg == 0.0 or cond is None) else cond)

    def _euler(self, prior, x1, order, times, dts, t_dtype, n_samples, sample_steps, use_ema, condition_cfg, mask_cfg, w_cfg,
               requires_grad, preserve_history, feed):
        """order[k]: schedule index of the k-th network evaluation; times[k] / dts[k]: what the backbone sees, step length."""
        log = {"sample_history": np.empty((n_samples, sample_steps + 1, *prior.shape)) if preserve_history else None}
        model = self.model_ema if use_ema else self.model
        xt = x1.clone()
        xt = xt * (1. - self.fix_mask) + prior * self.fix_mask
        if preserve_history:
            log["sample_history"][:, 0] = xt.cpu().numpy()
        with torch.set_grad_enabled(requires_grad):
            cond = model["condition"](condition_cfg, mask_cfg) if condition_cfg is not None else None
        fused = None
        if not preserve_history:
            from ..engine import dispatch
            from ..engine.plan import build_flow_plan
            plan = build_flow_plan(times, dts, self._integer_time)
            fused = dispatch.try_fused_sample(self, model, plan, xt, prior, cond, w_cfg if cond is not None else 0.0, 0.0,
                                              requires_grad, feed)
        if fused is not None:
            xt = fused
        else:
            for k, (tv, dt) in enumerate(zip(times, dts)):
                t = torch.full((n_samples,), tv, dtype=t_dtype, device=self.device)
                with torch.set_grad_enabled(requires_grad):
                    vel = self._velocity(model, xt, t, cond, w_cfg)
                xt = xt + dt * vel
                xt = xt * (1. - self.fix_mask) + prior * self.fix_mask
                if preserve_history:
                    log["sample_history"][:, sample_steps - order[k] + 1] = xt.cpu().numpy()
        if self.clip_pred:
            xt = xt.clip(self.x_min, self.x_max)
        return xt, log


class DiscreteRectifiedFlow(_RectifiedFlow):
    _integer_time = True

    def __init__(self, nn_diffusion: BaseNNDiffusion, nn_condition: Optional[BaseNNCondition] = None,
                 fix_mask=None, loss_weight=None, classifier=None, grad_clip_norm: Optional[float] = None,
                 ema_rate: float = 0.995, optim_params: Optional[dict] = None, diffusion_steps: int = 1000,
                 discretization: Union[str, Callable] = "uniform", x_max: Optional[torch.Tensor] = None,
                 x_min: Optional[torch.Tensor] = None, device: Union[torch.device, str] = "cpu"):
        super().__init__(nn_diffusion, nn_condition, fix_mask, loss_weight, classifier, grad_clip_norm, diffusion_steps,
                         ema_rate, optim_params, device)
        assert classifier is None, "Rectified Flow does not support classifier-guidance."
        self.x_max, self.x_min = x_max, x_min
        if isinstance(discretization, str):
            fn = SUPPORTED_DISCRETIZATIONS.get(discretization, SUPPORTED_DISCRETIZATIONS["uniform"])
            self.t_diffusion = fn(diffusion_steps, 0.).to(device)
        elif callable(discretization):
            self.t_diffusion = discretization(diffusion_steps, 0.).to(device)
        else:
            raise ValueError("discretization must be a callable or a string")

    def _training_times(self, n):
        t = torch.randint(self.diffusion_steps, (n,), device=self.device)
        return t, self.t_diffusion[t]

    @staticmethod
    def _interpolate(x0, x1, w):
        return w * x1 + (1 - w) * x0

    def sample(self, prior: torch.Tensor, x1: torch.Tensor = None, n_samples: int = 1, sample_steps: int = 5,
               sample_step_schedule: Union[str, Callable] = "uniform", use_ema: bool = True, temperature: float = 1.0,
               condition_cfg=None, mask_cfg=None, w_cfg: float = 0.0, condition_cg=None, w_cg: float = 0.0,
               diffusion_x_sampling_steps: int = 0, warm_start_reference: Optional[torch.Tensor] = None,
               warm_start_forward_level: float = 0.3, requires_grad: bool = False, preserve_history: bool = False,
               **kwargs):
        assert w_cg == 0.0 and condition_cg is None, "Rectified Flow does not support classifier-guidance."
        feed = _NoiseFeed(kwargs.get("noise", None))
        prior = prior.to(self.device)
        if isinstance(warm_start_reference, torch.Tensor):
            diffusion_steps = int(warm_start_forward_level * self.diffusion_steps)
            t_c = at_least_ndim(self.t_diffusion[diffusion_steps], prior.dim())
            x1 = feed.like(prior) * t_c + warm_start_reference * (1 - t_c)
        else:
            diffusion_steps = self.diffusion_steps
            if x1 is None:
                x1 = feed.like(prior) * temperature
            else:
                assert prior.shape == x1.shape, "prior and x1 must have the same shape"
        if isinstance(sample_step_schedule, str):
            if sample_step_schedule not in SUPPORTED_SAMPLING_STEP_SCHEDULE:
                raise ValueError(f"Sampling step schedule {sample_step_schedule} is not supported.")
            sched = SUPPORTED_SAMPLING_STEP_SCHEDULE[sample_step_schedule](diffusion_steps, sample_steps)
        elif callable(sample_step_schedule):
            sched = sample_step_schedule(diffusion_steps, sample_steps)
        else:
            raise ValueError("sample_step_schedule must be a callable or a string")
        order = list(reversed([1] * diffusion_x_sampling_steps + list(range(1, sample_steps + 1))))
        times = [int(sched[i]) for i in order]
        dts = [self.t_diffusion[sched[i]] - self.t_diffusion[sched[i - 1]] for i in order]
        return self._euler(prior, x1, order, times, dts, torch.long, n_samples, sample_steps, use_ema, condition_cfg, mask_cfg,
                           w_cfg, requires_grad, preserve_history, feed)


class ContinuousRectifiedFlow(_RectifiedFlow):
    def __init__(self, nn_diffusion: BaseNNDiffusion, nn_condition: Optional[BaseNNCondition] = None,
                 fix_mask=None, loss_weight=None, classifier=None, grad_clip_norm: Optional[float] = None,
                 ema_rate: float = 0.995, optim_params: Optional[dict] = None, x_max: Optional[torch.Tensor] = None,
                 x_min: Optional[torch.Tensor] = None, device: Union[torch.device, str] = "cpu"):
        super().__init__(nn_diffusion, nn_condition, fix_mask, loss_weight, classifier, grad_clip_norm, 0, ema_rate,
                         optim_params, device)
        assert classifier is None, "Rectified Flow does not support classifier-guidance."
        self.x_max, self.x_min = x_max, x_min

    def _training_times(self, n):
        t = torch.rand((n,), device=self.device)
        return t, t

    @staticmethod
    def _interpolate(x0, x1, w):
        return x0 + w * (x1 - x0)

    def sample(self, prior: torch.Tensor, x1: torch.Tensor = None, n_samples: int = 1, sample_steps: int = 5,
               sample_step_schedule: Union[str, Callable] = "uniform_continuous", use_ema: bool = True,
               temperature: float = 1.0, condition_cfg=None, mask_cfg=None, w_cfg: float = 0.0, condition_cg=None,
               w_cg: float = 0.0, diffusion_x_sampling_steps: int = 0,
               warm_start_reference: Optional[torch.Tensor] = None, warm_start_forward_level: float = 0.3,
               requires_grad: bool = False, preserve_history: bool = False, **kwargs):
        assert w_cg == 0.0 and condition_cg is None, "Rectified Flow does not support classifier-guidance."
        feed = _NoiseFeed(kwargs.get("noise", None))
        prior = prior.to(self.device)
        warm = isinstance(warm_start_reference, torch.Tensor)
        if warm:
            t_c = torch.ones_like(prior) * warm_start_forward_level
            x1 = feed.like(prior) * t_c + warm_start_reference * (1 - t_c)
        elif x1 is None:
            x1 = feed.like(prior) * temperature
        else:
            assert prior.shape == x1.shape, "prior and x1 must have the same shape"
        final_t = warm_start_forward_level if (warm and warm_start_forward_level > 0.) else 1.
        if isinstance(sample_step_schedule, str):
            if sample_step_schedule not in SUPPORTED_SAMPLING_STEP_SCHEDULE:
                raise ValueError(f"Sampling step schedule {sample_step_schedule} is not supported.")
            sched = SUPPORTED_SAMPLING_STEP_SCHEDULE[sample_step_schedule]([0., final_t], sample_steps)
        elif callable(sample_step_schedule):
            sched = sample_step_schedule([0., final_t], sample_steps)
        else:
            raise ValueError("sample_step_schedule must be a callable or a string")
        order = list(reversed([1] * diffusion_x_sampling_steps + list(range(1, sample_steps + 1))))
        times = [sched[i] for i in order]
        dts = [sched[i] - sched[i - 1] for i in order]
        return self._euler(prior, x1, order, times, dts, torch.float32, n_samples, sample_steps, use_ema, condition_cfg, mask_cfg,
                           w_cfg, requires_grad, preserve_history, feed)
