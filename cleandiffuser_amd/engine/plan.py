"""Step-plan compiler: turns (solver name, noise-schedule tables, step schedule) into a flat list of
per-step records that BOTH executors consume --

* the PyTorch executor in ``diffusion/diffusionsde.py`` (CPU, autograd, classifier guidance, unknown backbones)
* the fused gfx950 kernel (``csrc/cdx_unet2.hip``), which receives the same records as a ``cdx_step`` array.

Every scalar is evaluated exactly the way the reference evaluates it (0-dim fp32 torch ops in the same
association order, reference diffusionsde.py:514-589), then frozen to a Python float, so the only
per-element arithmetic left for the device is the affine update itself.

Update forms (P = clipped network output, eps/xth = noise/data prediction derived from P):
  DDPM   x <- k0*(x - k1*eps) + k2*eps  [+ k3*z]
  DDIM   x <- k0*((x - k1*eps)/k2) + k3*eps
  LINEAR x <- k0*x - k1*V [+ k2*z],   V in {eps, xth, D},  D = k3*xth - k4*xth_prev   (2M multistep)
"""
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import torch

SUPPORTED_SOLVERS = [
    "ddpm", "ddim",
    "ode_dpmsolver_1", "ode_dpmsolver++_1", "ode_dpmsolver++_2M",
    "sde_dpmsolver_1", "sde_dpmsolver++_1", "sde_dpmsolver++_2M"]

KIND_DDPM, KIND_DDIM, KIND_LINEAR, KIND_LEGACY_EPS, KIND_LEGACY_X0 = 0, 1, 2, 3, 4
V_EPS, V_XTHETA, V_MULTISTEP, V_MULTISTEP_EPS = 0, 1, 2, 3
F_MASK_PRED = 1


@dataclass
class Step:
    kind: int
    vsel: int
    index: int                 # i in the reference loop (position in the step schedule)
    t: float                   # schedule value fed to map_noise (int index or continuous time)
    alpha: float
    sigma: float
    k: Tuple[float, float, float, float, float]
    noise: bool = False        # consumes one fresh N(0, I) draw
    push: int = 0              # 1: stores xth (2: eps) for the next multistep update; EDM: stores slope and state
    flags: int = 0             # F_MASK_PRED: legacy DPMSolver applies the fix-mask to the prediction


@dataclass
class SamplePlan:
    solver: str
    steps: List[Step] = field(default_factory=list)
    t_is_integer: bool = True
    clip_each_step: bool = True            # False: the solver has no per-step clipping (rectified flow)
    network_predicts_noise: object = None  # None: ask the solver object; True/False: the plan decides (rectified flow)

    @property
    def n_noise(self) -> int:
        return sum(1 for s in self.steps if s.noise)


def cached(plan: "SamplePlan", key, make):
    """Per-plan memo for the device-side images of a plan (timestep vector, packed step records): plans are immutable once
    built and the solvers keep them across calls, so a steady-state ``sample()`` issues no host-to-device copy for them --
    a pageable H2D copy is stream-ordered behind the previous call's kernel and would stall the host on it."""
    memo = plan.__dict__.setdefault("_memo", {})
    if key not in memo:
        memo[key] = make()
    return memo[key]


def _f(x) -> float:
    return float(x.item()) if isinstance(x, torch.Tensor) else float(x)


def vp_tables(alphas: torch.Tensor, sigmas: torch.Tensor):
    """logSNR increments h_i and posterior stds (reference diffusionsde.py:516-520)."""
    alphas, sigmas = alphas.detach().float().cpu(), sigmas.detach().float().cpu()
    log_snr = torch.log(alphas / sigmas)
    hs = torch.zeros_like(log_snr)
    hs[1:] = log_snr[:-1] - log_snr[1:]
    stds = torch.zeros_like(log_snr)
    stds[1:] = sigmas[:-1] / sigmas[1:] * (1 - (alphas[1:] / alphas[:-1]) ** 2).sqrt()
    return alphas, sigmas, hs, stds


def build_vp_plan(solver: str, alphas: torch.Tensor, sigmas: torch.Tensor, schedule: Sequence,
                  sample_steps: int, diffusion_x_sampling_steps: int = 0, t_is_integer: bool = True) -> SamplePlan:
    assert solver in SUPPORTED_SOLVERS, f"Solver {solver} is not supported."
    a, s, hs, stds = vp_tables(alphas, sigmas)
    order = list(reversed([1] * diffusion_x_sampling_steps + list(range(1, sample_steps + 1))))
    plan = SamplePlan(solver=solver, t_is_integer=t_is_integer)
    n_pushed = 0
    for i in order:
        tval = schedule[i]
        tval = int(tval) if t_is_integer else _f(tval)
        zero = 0.0
        if solver == "ddpm":
            st = Step(KIND_DDPM, V_EPS, i, tval, _f(a[i]), _f(s[i]),
                      (_f(a[i - 1] / a[i]), _f(s[i]), _f((s[i - 1] ** 2 - stds[i] ** 2 + 1e-8).sqrt()),
                       _f(stds[i]), zero), noise=(i > 1))
        elif solver == "ddim":
            st = Step(KIND_DDIM, V_EPS, i, tval, _f(a[i]), _f(s[i]),
                      (_f(a[i - 1]), _f(s[i]), _f(a[i]), _f(s[i - 1]), zero))
        elif solver == "ode_dpmsolver_1":
            st = Step(KIND_LINEAR, V_EPS, i, tval, _f(a[i]), _f(s[i]),
                      (_f(a[i - 1] / a[i]), _f(s[i - 1] * torch.expm1(hs[i])), zero, zero, zero))
        elif solver == "sde_dpmsolver_1":
            st = Step(KIND_LINEAR, V_EPS, i, tval, _f(a[i]), _f(s[i]),
                      (_f(a[i - 1] / a[i]), _f(2 * s[i - 1] * torch.expm1(hs[i])),
                       _f(s[i - 1] * torch.expm1(2 * hs[i]).sqrt()), zero, zero), noise=True)
        else:
            multistep = solver.endswith("2M")
            stochastic = solver.startswith("sde")
            if stochastic:
                k0 = _f((s[i - 1] / s[i]) * (-hs[i]).exp())
                k1 = _f(a[i - 1] * torch.expm1(-2 * hs[i]))
                k2 = _f(s[i - 1] * (-torch.expm1(-2 * hs[i])).sqrt())
            else:
                k0 = _f(s[i - 1] / s[i])
                k1 = _f(a[i - 1] * torch.expm1(-hs[i]))
                k2 = zero
            k3 = k4 = zero
            vsel = V_XTHETA
            # the reference tests `i < sample_steps`; with at least one stored xth that is "not the first step"
            if multistep and i < sample_steps and n_pushed >= 1:
                r = hs[i + 1] / hs[i]
                k3, k4 = _f(1 + 0.5 / r), _f(0.5 / r)
                vsel = V_MULTISTEP
            st = Step(KIND_LINEAR, vsel, i, tval, _f(a[i]), _f(s[i]), (k0, k1, k2, k3, k4),
                      noise=stochastic, push=multistep)
            n_pushed += 1 if multistep else 0
        plan.steps.append(st)
    return plan


def build_legacy_ddpm_plan(beta: torch.Tensor, alpha: torch.Tensor, bar_alpha: torch.Tensor, predict_noise: bool,
                           extra_sample_steps: int = 0) -> SamplePlan:
    """Legacy ``DDPM`` class (reference diffusion/ddpm.py:212-241 and the Diffusion-X tail :321-343): ancestral steps
    t = T-1 .. 0 over ALL diffusion steps, then `extra_sample_steps` noise-free repeats of t = 0.
      eps:  x <- 1/sqrt(alpha_t) * (x - beta_t/sqrt(1-abar_t) * P)                     [+ sqrt(beta_t (1-abar_{t-1})/(1-abar_t)) z]
      x0 :  x <- 1/(1-abar_t) * (sqrt(alpha_t)(1-abar_{t-1}) x + beta_t sqrt(abar_{t-1}) P)   [+ same noise]
    `alpha`/`sigma` of a record carry sqrt(abar_t), sqrt(1-abar_t): the clip bounds use them (ddpm.py:153-160)."""
    beta, alpha, bar = (v.detach().float().cpu() for v in (beta, alpha, bar_alpha))
    T = beta.shape[0]
    plan = SamplePlan(solver="legacy_ddpm", t_is_integer=True)
    one = torch.tensor(1.0)
    order = list(range(T - 1, -1, -1)) + [0] * extra_sample_steps
    for pos, t in enumerate(order):
        bar_prev = bar[t - 1] if t > 0 else one
        if predict_noise:
            k = (_f(1 / alpha[t].sqrt()), _f(beta[t] / (1 - bar[t]).sqrt()), 0.0)
            kind = KIND_LEGACY_EPS
        else:
            k = (_f(1 / (1 - bar[t])), _f(alpha[t].sqrt() * (1 - bar_prev)), _f(beta[t] * bar_prev.sqrt()))
            kind = KIND_LEGACY_X0
        std = _f((beta[t] * (1 - bar_prev) / (1 - bar[t])).sqrt())
        plan.steps.append(Step(kind, V_EPS, t, int(t), _f(bar[t].sqrt()), _f((1 - bar[t]).sqrt()),
                               (k[0], k[1], k[2], std, 0.0), noise=(t != 0 and pos < T)))
    return plan


KIND_EDM_EULER, KIND_EDM_HEUN = 5, 6


def build_edm_plan(sigma_data: float, sigma_min: float, top_sigma: float, rho: float, sample_steps: int, solver: str,
                   diffusion_x_sampling_steps: int = 0) -> SamplePlan:
    """``ContinuousEDM.sample`` (reference diffusion/newedm.py:373-401): Karras sigma ladder, Euler predictor and (Heun,
    i > 1) trapezoid corrector.  One record per network evaluation; ``t`` is c_noise = ln(sigma)/4 as the backbone's
    ``map_noise`` sees it, ``alpha`` carries c_in, ``k`` = (c_skip, c_out, sigma, dt, 0):
        euler  D = clip(k0 x + k1 F);  s = (x - D)/k2;  x <- x - k3 s
        heun   D = clip(k0 x' + k1 F); s' = (x' - D)/k2; x <- x_old - k3 (s + s')/2     (x' = Euler result)
    Scalars are produced by the same 0-dim fp32 torch expressions as the reference."""
    inv_rho = 1 / rho
    ramp = torch.arange(sample_steps + 1) / sample_steps
    sigmas = (sigma_min ** inv_rho + ramp * (top_sigma ** inv_rho - sigma_min ** inv_rho)) ** rho
    sd2 = sigma_data ** 2

    def pre(sig):                                       # sig: 0-dim fp32 tensor, as `t` reaches D() in the reference
        return (_f(sd2 / (sd2 + sig ** 2)), _f(sig * sigma_data / (sd2 + sig ** 2).sqrt()),
                _f(1 / (sd2 + sig ** 2).sqrt()), _f(0.25 * sig.log()))

    plan = SamplePlan(solver="edm_" + solver, t_is_integer=False)
    for i in reversed([1] * diffusion_x_sampling_steps + list(range(1, sample_steps + 1))):
        sig, sig_prev = sigmas[i], sigmas[i - 1]
        dt = _f(sig - sig_prev)
        skip, out, cin, cnoise = pre(sig)
        corrector = solver == "heun" and i > 1
        plan.steps.append(Step(KIND_EDM_EULER, V_EPS, i, cnoise, cin, _f(sig), (skip, out, _f(sig), dt, 0.0),
                               push=corrector))
        if corrector:
            sig2 = sig / sig * sig_prev                  # reference: t / sigmas[i] * sigmas[i-1]
            skip, out, cin, cnoise = pre(sig2)
            plan.steps.append(Step(KIND_EDM_HEUN, V_EPS, i, cnoise, cin, _f(sig2), (skip, out, _f(sig_prev), dt, 0.0)))
    return plan


# ------------------------------------------------------------------------------------------------ #
# legacy classes the dp_* / dbc_* pipelines import: DPMSolver, EDM                                     #
# ------------------------------------------------------------------------------------------------ #
LEGACY_DPM_SAMPLERS = {          # name -> (prediction the update consumes is eps?, order)   (reference dpmsolver.py:13-49)
    "ode_dpm_1": (True, 1), "ddim": (True, 1), "sde_dpm_1": (True, 1), "ode_dpmpp_1": (False, 1), "sde_dpmpp_1": (False, 1),
    "ode_dpm_2": (True, 2), "sde_dpm_2": (True, 2), "ode_dpmpp_2": (False, 2), "sde_dpmpp_2": (False, 2)}


def build_legacy_dpmsolver_plan(t: torch.Tensor, alphas: torch.Tensor, sigmas: torch.Tensor, sampler: str,
                                sample_steps: int, extra_sample_steps: int = 0) -> SamplePlan:
    """Legacy ``DPMSolver.sample / sample_x`` (reference diffusion/dpmsolver.py:66-89 one-step estimates, :478-520 loop,
    :600-616 Diffusion-X tail).  Step i = 1..S evaluates the network at t[i-1] and hops to t[i]:
        ode_dpm / ddim   x <- a_i/a_{i-1} x - s_i expm1(h_i) V                                 V eps-type
        sde_dpm          x <- a_i/a_{i-1} x - 2 s_i expm1(h_i) V + s_i sqrt(expm1(2 h_i)) z
        ode_dpmpp        x <- s_i/s_{i-1} x - a_i expm1(-h_i) V                                V x-type
        sde_dpmpp        x <- s_i/s_{i-1} e^{-h_i} x - a_i expm1(-2 h_i) V + s_i sqrt(-expm1(-2 h_i)) z
    second order (i > 1): V = (1 + 1/(2r)) P_i - 1/(2r) P_{i-1}, r = h_{i-1}/h_i, P the masked prediction (flag MASK_PRED).
    `alpha`/`sigma` of a record are those of t[i-1] (conversion and clipping happen there)."""
    eps_type, order = LEGACY_DPM_SAMPLERS[sampler]
    t, alphas, sigmas = (v.detach().float().cpu() for v in (t, alphas, sigmas))
    log_snr = (alphas / sigmas).log()
    h = torch.zeros_like(log_snr)
    h[1:] = log_snr[1:] - log_snr[:-1]
    family = "ode_dpm" if sampler == "ddim" else sampler[:-2]
    plan = SamplePlan(solver="legacy_" + sampler, t_is_integer=False)
    order_of = list(range(1, sample_steps + 1)) + ([sample_steps] * extra_sample_steps if order == 1 else [])
    for pos, i in enumerate(order_of):
        if family == "ode_dpm":
            c = (_f(alphas[i] / alphas[i - 1]), _f(sigmas[i] * torch.expm1(h[i])), 0.0)
        elif family == "sde_dpm":
            c = (_f(alphas[i] / alphas[i - 1]), _f(2. * sigmas[i] * torch.expm1(h[i])),
                 _f(sigmas[i] * torch.expm1(2. * h[i]).sqrt()))
        elif family == "ode_dpmpp":
            c = (_f(sigmas[i] / sigmas[i - 1]), _f(alphas[i] * torch.expm1(-h[i])), 0.0)
        else:
            c = (_f(sigmas[i] / sigmas[i - 1] * (-h[i]).exp()), _f(alphas[i] * torch.expm1(-2. * h[i])),
                 _f(sigmas[i] * (-1. * torch.expm1(-2. * h[i])).sqrt()))
        k3 = k4 = 0.0
        vsel = V_EPS if eps_type else V_XTHETA
        if order == 2 and i > 1 and pos < sample_steps:
            r = h[i - 1] / h[i]
            k3, k4 = _f(1 + 0.5 / r), _f(0.5 / r)
            vsel = V_MULTISTEP_EPS if eps_type else V_MULTISTEP
        plan.steps.append(Step(KIND_LINEAR, vsel, i, _f(t[i - 1]), _f(alphas[i - 1]), _f(sigmas[i - 1]),
                               (c[0], c[1], c[2], k3, k4), noise=family.startswith("sde"),
                               push=(2 if eps_type else 1) if order == 2 else 0, flags=F_MASK_PRED))
    return plan


def build_legacy_edm_plan(sigma_data: float, sigma_s: torch.Tensor, solver: str, extra_sample_steps: int = 0) -> SamplePlan:
    """Legacy ``EDM.sample / sample_x`` (reference diffusion/edm.py:118-160 dot_x, :252-268 loop, :330-341 tail) for the
    ``EDM`` class proper (scale_s == 1, t_s == sigma_s): Euler  x <- x - (x - D)/sigma_i (sigma_i - sigma_{i+1}), Heun corrector
    when i != N-1 and sigma_{i+1} > 0.005, `extra_sample_steps` repeats of the last Euler step.  No clipping, no temperature."""
    sig = sigma_s.detach().float().cpu()
    n = sig.shape[0] - 1
    sd2 = sigma_data ** 2

    def rec(kind, i, dt, push):
        s = sig[i]
        return Step(kind, V_EPS, i, _f(0.25 * s.log()), _f(1 / (sd2 + s ** 2).sqrt()), _f(s),
                    (_f(sd2 / (sd2 + s ** 2)), _f(s * sigma_data / (sd2 + s ** 2).sqrt()), _f(s), dt, 0.0), push=push)

    plan = SamplePlan(solver="legacy_edm_" + solver, t_is_integer=False)
    for i in range(n):
        dt = _f(sig[i] - sig[i + 1])
        corrector = solver == "heun" and i != n - 1 and bool(sig[i + 1] > 0.005)
        plan.steps.append(rec(KIND_EDM_EULER, i, dt, int(corrector)))
        if corrector:
            plan.steps.append(rec(KIND_EDM_HEUN, i + 1, dt, 0))
    for _ in range(extra_sample_steps):
        plan.steps.append(rec(KIND_EDM_EULER, n - 1, _f(sig[n - 1] - sig[n]), 0))
    return plan


KIND_CONSISTENCY = 7


def build_flow_plan(times, dts, integer_time: bool) -> SamplePlan:
    """Rectified-flow Euler steps (reference diffusion/rectifiedflow.py:318-334, :612-628): x <- x + dt v(x, t).  As ``linear``
    records: alpha = sigma = 1 and "predict noise" so that V is the raw network output, k0 = 1, k1 = -dt.  The per-step
    clipping of the VP solvers does not exist here (``clip_each_step = False``); the class clips once at the end."""
    plan = SamplePlan(solver="rectified_flow_euler", t_is_integer=integer_time)
    plan.clip_each_step = False
    plan.network_predicts_noise = True
    for k, (t, dt) in enumerate(zip(times, dts)):
        plan.steps.append(Step(KIND_LINEAR, V_EPS, k, int(t) if integer_time else _f(t), 1.0, 1.0, (1.0, -_f(dt), 0.0, 0.0, 0.0)))
    return plan


def build_consistency_plan(sigma_data: float, sigma_min: float, sigmas: torch.Tensor, levels) -> SamplePlan:
    """Multistep consistency sampling (reference diffusion/consistency_model.py:412-427): record j evaluates
    f(x, sigma_{levels[j]}) = clip(c_skip x + c_out F(c_in x, ln(sigma)/4)), applies the fix-mask, and -- unless it is the last
    record -- re-noises for the next level: x <- f + sqrt(sigma_next^2 - sigma_min^2) z.  Kind 7: k = (c_skip, c_out, -, noise
    scale), ``alpha`` = c_in, boundary-condition preconditioning c_skip = sd^2 / (sd^2 + (s - smin)^2), c_out = (s - smin) sd / sqrt(sd^2 + s^2)."""
    sig = sigmas.detach().float().cpu()
    sd2 = sigma_data ** 2
    plan = SamplePlan(solver="consistency", t_is_integer=False)
    for j, i in enumerate(levels):
        s = sig[i]
        last = j == len(levels) - 1
        renoise = 0.0 if last else _f((sig[levels[j + 1]] ** 2 - sigma_min ** 2).sqrt())
        plan.steps.append(Step(KIND_CONSISTENCY, V_EPS, i, _f(0.25 * s.log()), _f(1 / (sd2 + s ** 2).sqrt()), _f(s),
                               (_f(sd2 / (sd2 + (s - sigma_min) ** 2)), _f((s - sigma_min) * sigma_data / (sd2 + s ** 2).sqrt()),
                                1.0, renoise, 0.0), noise=not last))
    return plan
