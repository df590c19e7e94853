"""``ContinuousConsistencyModel`` -- consistency training / distillation and multistep consistency sampling.

Contract: reference diffusion/consistency_model.py:15-48 (erf polynomial, property diff, pseudo-Huber), :51-87 (curriculum
logger: N(k) doubling schedule, Karras sigma ladder, lognormal bin probabilities), :90-261 (ctor, preconditioning with the
``sigma - sigma_min`` boundary condition, ``f``), :264-363 (distillation / training losses, update), :366-428 (``sample``:
x <- f(x, sigma_S); then for i = S-1..1: x <- f(x + sqrt(sigma_i^2 - sigma_min^2) z, sigma_i), fix-mask after every f).

Execution: on a ROCm device ``sample`` is one native call (``engine/plan.py:build_consistency_plan``, step kind 7 on top of the
EDM input scaling); otherwise the PyTorch loop below.
"""
from typing import Callable, List, Optional, Union

import numpy as np
import torch
import torch.nn as nn

from ..nn_condition import BaseNNCondition
from ..nn_diffusion import BaseNNDiffusion
from ..utils import at_least_ndim
from .basic import DiffusionModel
from .diffusionsde import _NoiseFeed
from .newedm import ContinuousEDM


def erf(x):
    """Abramowitz-Stegun 7.1.26 polynomial (numpy), as the reference evaluates the lognormal bin masses."""
    coef = (0.254829592, -0.284496736, 1.421413741, -1.453152027, 1.061405429)
    sign, x = np.sign(x), np.abs(x)
    t = 1.0 / (1.0 + 0.3275911 * x)
    poly = ((((coef[4] * t + coef[3]) * t) + coef[2]) * t + coef[1]) * t + coef[0]
    return sign * (1.0 - poly * t * np.exp(-x * x))


def compare_properties(obj1, obj2, properties: List[str]):
    diff = []
    for name in properties:
        a, b = getattr(obj1, name), getattr(obj2, name)
        if isinstance(a, torch.Tensor):
            same = torch.allclose(a, b)
        elif isinstance(a, np.ndarray):
            same = np.allclose(a, b)
        else:
            same = a == b
        if not same:
            diff.append(name)
    return diff


def pseudo_huber_loss(source: torch.Tensor, target: torch.Tensor, c: float = 0.0):
    return ((source - target) ** 2 + c ** 2).sqrt() - c


def _karras(sigma_min, sigma_max, rho, ramp):
    return (sigma_min ** (1 / rho) + ramp * (sigma_max ** (1 / rho) - sigma_min ** (1 / rho))) ** rho


class CMCurriculumLogger:
    def __init__(self, s0: int = 10, s1: int = 1280, curriculum_cycle: int = 100_000, sigma_min: float = 0.002,
                 sigma_max: float = 80., rho: float = 7., P_mean: float = -1.1, P_std: float = 2.0):
        self.Kprime = np.ceil(curriculum_cycle / (np.log2(np.ceil(s1 / s0)) + 1))
        self.Nk, self.s0, self.s1, self.curriculum_cycle = s0, s0, s1, curriculum_cycle
        self.sigma_min, self.sigma_max, self.rho, self.P_mean, self.P_std = sigma_min, sigma_max, rho, P_mean, P_std
        self.ceil_k_div_Kprime, self.k = None, None
        self.update_k(0)

    def update_k(self, k):
        self.k = k
        stage = np.ceil(k / self.Kprime)
        if stage != self.ceil_k_div_Kprime:
            self.ceil_k_div_Kprime = stage
            self.Nk = int(min(self.s0 * (2 ** stage), self.s1))
            self.sigmas = _karras(self.sigma_min, self.sigma_max, self.rho, np.arange(self.Nk + 1, dtype=np.float32) / self.Nk)
            z = (np.log(self.sigmas) - self.P_mean) / (self.P_std * (2 ** 0.5))
            mass = erf(z[1:]) - erf(z[:-1])
            self.p_sigmas = mass / mass.sum()

    def incremental_update_k(self):
        self.update_k(self.k + 1)

    @property
    def curriculum_process(self):
        return (self.k % self.curriculum_cycle) / self.curriculum_cycle


class ContinuousConsistencyModel(DiffusionModel):
    def __init__(self, nn_diffusion: BaseNNDiffusion, nn_condition: Optional[BaseNNCondition] = None,
                 fix_mask=None, loss_weight=None, classifier=None, grad_clip_norm: Optional[float] = None,
                 ema_rate: float = 0.9999, optim_params: Optional[dict] = None, s0: int = 10, s1: int = 1280,
                 data_dim: int = None, P_mean: float = -1.1, P_std: float = 2.0, sigma_min: float = 0.002,
                 sigma_max: float = 80., sigma_data: float = 0.5, rho: float = 7.0, curriculum_cycle: int = 100_000,
                 x_max: Optional[torch.Tensor] = None, x_min: Optional[torch.Tensor] = None,
                 device: Union[torch.device, str] = "cpu"):
        super().__init__(nn_diffusion, nn_condition, fix_mask, loss_weight, classifier, grad_clip_norm, 0, ema_rate,
                         optim_params, device)
        self.cur_logger = CMCurriculumLogger(s0, s1, curriculum_cycle, sigma_min, sigma_max, rho, P_mean, P_std)
        self.pseudo_huber_constant = 0.01 if data_dim is None else 0.00054 * np.sqrt(data_dim)
        self.rho, self.sigma_data, self.sigma_max, self.sigma_min = rho, sigma_data, sigma_max, sigma_min
        self.x_max = x_max.to(device) if isinstance(x_max, torch.Tensor) else x_max
        self.x_min = x_min.to(device) if isinstance(x_min, torch.Tensor) else x_min
        self.edm = None
        self.distillation_sigmas, self.distillation_N = None, None

    def prepare_distillation(self, edm: ContinuousEDM, distillation_N: int = 18):
        shared = ["sigma_data", "sigma_max", "sigma_min", "rho", "x_max", "x_min", "fix_mask", "loss_weight", "device"]
        diff = compare_properties(self, edm, shared)
        if diff:
            raise ValueError(f"Properties {diff} are different between the EDM and the Consistency Model.")
        self.edm = edm
        self.model.load_state_dict(edm.model.state_dict())
        self.model_ema.load_state_dict(edm.model_ema.state_dict())
        self.distillation_N = distillation_N
        self.distillation_sigmas = self.training_noise_schedule(distillation_N)

    @property
    def supported_solvers(self):
        return ["none"]

    @property
    def clip_pred(self):
        return (self.x_max is not None) or (self.x_min is not None)

    def training_noise_schedule(self, N):
        return torch.tensor(_karras(self.sigma_min, self.sigma_max, self.rho, np.arange(N + 1) / N), device=self.device,
                            dtype=torch.float32)

    # preconditioning with f(x, sigma_min) = x
    def c_skip(self, sigma):
        return self.sigma_data ** 2 / (self.sigma_data ** 2 + (sigma - self.sigma_min) ** 2)

    def c_out(self, sigma):
        return (sigma - self.sigma_min) * self.sigma_data / (self.sigma_data ** 2 + sigma ** 2).sqrt()

    def c_in(self, sigma):
        return 1 / (self.sigma_data ** 2 + sigma ** 2).sqrt()

    def c_noise(self, sigma):
        return 0.25 * sigma.log()

    def f(self, x, t, condition=None, model=None):
        model = self.model if model is None else model
        skip, out, inn = (at_least_ndim(c(t), x.dim()) for c in (self.c_skip, self.c_out, self.c_in))
        pred = skip * x + out * model["diffusion"](inn * x, self.c_noise(t), condition)
        return pred.clip(self.x_min, self.x_max) if self.clip_pred else pred

    # ------------------------------------ training -------------------------------------------- #
    def distillation_loss(self, x0, condition=None):
        assert self.edm is not None, "Please call `prepare_distillation` before distillation."
        idx = torch.randint(self.distillation_N, (x0.shape[0],), device=self.device)
        t_m, t_n = self.distillation_sigmas[idx + 1], self.distillation_sigmas[idx]
        x_m, t_m, _ = self.edm.add_noise(x0, t_m, None)
        with torch.no_grad():
            teacher_cond = self.edm.model_ema["condition"](condition) if condition is not None else None
            pred, _ = self.edm.guided_sampling(x_m, t_m, None, self.edm.model_ema, teacher_cond, 1.0, None, 0.0, False)
            slope = (x_m - pred) / at_least_ndim(t_m, x_m.dim())
            x_n = x_m - slope * at_least_ndim(t_m - t_n, x_m.dim())
            x_n = x_n * (1. - self.fix_mask) + x0 * self.fix_mask
        cond = self.model["condition"](condition) if condition is not None else None
        pred_m = self.f(x_m, t_m, cond, self.model)
        with torch.no_grad():
            cond_ema = self.model_ema["condition"](condition) if condition is not None else None
            pred_n = self.f(x_n, t_n, cond_ema, self.model_ema)
        loss = ((pred_n - pred_m) ** 2) * (1 - self.fix_mask) * self.loss_weight * \
            at_least_ndim((1 / (t_m - t_n)), pred_n.dim())
        return loss.mean(), None

    def training_loss(self, x0, condition=None):
        idx = np.random.choice(self.cur_logger.Nk, size=x0.shape[0], p=self.cur_logger.p_sigmas)
        sigma_n = torch.tensor(self.cur_logger.sigmas[idx], device=self.device)
        sigma_m = torch.tensor(self.cur_logger.sigmas[idx + 1], device=self.device)
        eps = torch.randn_like(x0)
        x_n = x0 + at_least_ndim(sigma_n, x0.dim()) * eps
        x_m = x0 + at_least_ndim(sigma_m, x0.dim()) * eps
        cond = self.model["condition"](condition) if condition is not None else None
        pred_m = self.f(x_m, sigma_m, cond, self.model)
        with torch.no_grad():
            pred_n = self.f(x_n, sigma_n, None if cond is None else cond.detach(), self.model)
        unweighted = pseudo_huber_loss(pred_m, pred_n, self.pseudo_huber_constant) * (1 - self.fix_mask) * self.loss_weight
        weight = at_least_ndim(1 / (sigma_m - sigma_n), x0.dim())
        return (unweighted * weight).mean(), unweighted.mean().item()

    def update(self, x0, condition=None, update_ema=True, loss_type="training", **kwargs):
        if loss_type == "training":
            loss, unweighted = self.training_loss(x0, condition)
        elif loss_type == "distillation":
            loss, unweighted = self.distillation_loss(x0, condition)
        else:
            raise ValueError(f"Unknown loss type: {loss_type}")
        # (the denoiser's forward ran on the library's training nodes -- its forward() routes there under autograd --; inside this backward
        #  they add the parameter gradients straight into .grad and issue the weight-gradient products as batched launches.  The step
        #  itself is not captured as a HIP graph: the training loss draws from numpy and reports a Python float per call)
        from ..engine import train
        with train.grads_in_place(self.model.parameters()):
            loss.backward()
        grad_norm = self._apply_gradients(update_ema)
        if loss_type == "training":
            self.cur_logger.incremental_update_k()
        return {"loss": loss.item(), "grad_norm": grad_norm, "unweighted_loss": unweighted}

    # ------------------------------------ sampling -------------------------------------------- #
    def sample(self, prior: torch.Tensor, solver: str = "none", n_samples: int = 1, sample_steps: int = 5,
               sample_step_schedule: Union[str, Callable] = "uniform", use_ema: bool = True, temperature: float = 1.0,
               condition_cfg=None, mask_cfg=None, w_cfg: float = 0.0, condition_cg=None, w_cg: float = 0.0,
               diffusion_x_sampling_steps: int = 0, warm_start_reference: Optional[torch.Tensor] = None,
               warm_start_forward_level: float = 0.3, requires_grad: bool = False, preserve_history: bool = False,
               **kwargs):
        assert w_cg == 0.0 and condition_cg is None, "Consistency Distillation does not support classifier guidance."
        feed = _NoiseFeed(kwargs.get("noise", None))
        log = {"sample_history": np.empty((n_samples, sample_steps + 1, *prior.shape)) if preserve_history else None}
        model = self.model_ema if use_ema else self.model
        prior = prior.to(self.device)
        xt = feed.like(prior) * self.sigma_max * temperature
        xt = xt * (1. - self.fix_mask) + prior * self.fix_mask
        if preserve_history:
            log["sample_history"][:, 0] = xt.cpu().numpy()
        with torch.set_grad_enabled(requires_grad):
            cond = model["condition"](condition_cfg, mask_cfg) if condition_cfg is not None else None
        sigmas = _karras(self.sigma_min, self.sigma_max, self.rho,
                         torch.arange(sample_steps + 1, device=self.device) / sample_steps)
        levels = [sample_steps] + list(reversed([1] * diffusion_x_sampling_steps + list(range(1, sample_steps))))

        if not preserve_history:
            from ..engine import dispatch
            from ..engine.plan import build_consistency_plan
            plan = build_consistency_plan(self.sigma_data, self.sigma_min, sigmas, levels)
            fused = dispatch.try_fused_edm(self, model, plan, xt, prior, cond, 1.0 if cond is not None else 0.0, 0.0,
                                           requires_grad, feed)
            if fused is not None:
                return fused, log

        pred = None
        feed.reserve(xt, len(levels) - 1)              # (as the whole-loop executor draws them: _NoiseFeed.reserve)
        for k, i in enumerate(levels):
            t = torch.full((n_samples,), sigmas[i], dtype=torch.float32, device=self.device)
            if k > 0:
                xt = pred + (at_least_ndim(t, xt.dim()) ** 2 - self.sigma_min ** 2).sqrt() * feed.like(xt)
            with torch.set_grad_enabled(requires_grad):
                pred = self.f(xt, t, cond, model)
            pred = pred * (1. - self.fix_mask) + prior * self.fix_mask
        return pred, log
