"""CPU restatement of the reference algorithm for the sampling hot path -- TEST INFRASTRUCTURE (oracle/__init__.py).

Functional, ``state_dict``-driven, no classes from cleandiffuser_amd: the checker must not share code with the
thing it checks.  The arithmetic lives in PyTorch ATen (the reference's own third-party dependency, pinned
``torch>1.0.0,<2.3.0`` in its pyproject.toml:33; this container has 2.10.0+rocm7.0), so the port calls the same ATen
CPU ops at the reference's call sites:

* ``janner_forward``     <- cleandiffuser/nn_diffusion/jannerunet.py:154-201 (ResidualBlock :66-69, Down/Upsample :24,33,
                            GroupNorm1d utils/building_blocks.py:60-76, PositionalEmbedding utils/utils.py:248-263)
* ``vp_sample``          <- cleandiffuser/diffusion/diffusionsde.py:478-606 (discrete) / :743-952 (continuous):
                            init :493-494, tables :514-520, loop :525-592, CFG :175-206, clip :208-223

Pinned against tests/golden/*.npz (outputs of the real reference, oracle/gen_golden.py) by tests/test_oracle_ports.py.
"""
import math
from typing import List, Optional

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ #
# backbone                                                                                           #
# ------------------------------------------------------------------------------------------------ #
def positional_embedding(t: torch.Tensor, dim: int, max_positions: int = 10000) -> torch.Tensor:
    freqs = torch.arange(0, dim // 2, dtype=torch.float32) / (dim // 2)
    freqs = (1 / max_positions) ** freqs
    ang = t.ger(freqs.to(t.dtype))                 # integer t truncates the frequencies (quirk Q1)
    return torch.cat([ang.cos(), ang.sin()], dim=1)


def _gn(x, sd, prefix, c):
    groups = min(8, c // 4)
    return F.group_norm(x.unsqueeze(2), groups, sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5).squeeze(2)


def _cna(x, sd, prefix, k):
    w = sd[prefix + ".0.weight"]
    y = F.conv1d(x, w, sd[prefix + ".0.bias"], padding=k // 2)
    return F.mish(_gn(y, sd, prefix + ".1", w.shape[0]))


def _resblock(x, emb, sd, prefix, k):
    h = _cna(x, sd, prefix + ".conv1", k)
    h = h + F.linear(F.mish(emb), sd[prefix + ".emb_mlp.1.weight"], sd[prefix + ".emb_mlp.1.bias"]).unsqueeze(-1)
    h = _cna(h, sd, prefix + ".conv2", k)
    if prefix + ".residual_conv.weight" in sd:
        res = F.conv1d(x, sd[prefix + ".residual_conv.weight"], sd[prefix + ".residual_conv.bias"])
    else:
        res = x
    return h + res


def janner_forward(sd, x, t, cond, *, emb_dim, kernel_size, n_levels):
    """x (b, H, D), t (b,), cond (b, emb_dim)|None -> (b, H, D).  `sd` = backbone state_dict (reference names)."""
    k = kernel_size
    x = x.permute(0, 2, 1)
    emb = positional_embedding(t, emb_dim)
    emb = emb + (cond if cond is not None else torch.zeros_like(emb))
    emb = F.linear(F.mish(F.linear(emb, sd["map_emb.0.weight"], sd["map_emb.0.bias"])),
                   sd["map_emb.2.weight"], sd["map_emb.2.bias"])
    skips = []
    for i in range(n_levels):
        x = _resblock(x, emb, sd, f"downs.{i}.0", k)
        x = _resblock(x, emb, sd, f"downs.{i}.1", k)
        skips.append(x)
        if f"downs.{i}.3.conv.weight" in sd:
            x = F.conv1d(x, sd[f"downs.{i}.3.conv.weight"], sd[f"downs.{i}.3.conv.bias"], stride=2, padding=1)
    x = _resblock(x, emb, sd, "mid_block1", k)
    x = _resblock(x, emb, sd, "mid_block2", k)
    for i in range(n_levels - 1):
        x = torch.cat([x, skips.pop()], dim=1)
        x = _resblock(x, emb, sd, f"ups.{i}.0", k)
        x = _resblock(x, emb, sd, f"ups.{i}.1", k)
        x = F.conv_transpose1d(x, sd[f"ups.{i}.3.conv.weight"], sd[f"ups.{i}.3.conv.bias"], stride=2, padding=1)
    y = F.conv1d(x, sd["final_conv.0.weight"], sd["final_conv.0.bias"], padding=2)
    y = F.mish(_gn(y, sd, "final_conv.1", y.shape[1]))
    y = F.conv1d(y, sd["final_conv.3.weight"], sd["final_conv.3.bias"])
    return y.permute(0, 2, 1)


# ------------------------------------------------------------------------------------------------ #
# schedules                                                                                          #
# ------------------------------------------------------------------------------------------------ #
def cosine_schedule(t, s=0.008):
    alpha = (np.pi / 2.0 * (t.clip(0., 0.9946) + s) / (1 + s)).cos() / np.cos(np.pi / 2.0 * s / (1 + s))
    return alpha, (1.0 - alpha ** 2).sqrt()


def linear_schedule(t, beta0=0.1, beta1=20.0):
    alpha = (-(beta1 - beta0) / 4.0 * (t ** 2) - beta0 / 2.0 * t).exp()
    return alpha, (1.0 - alpha ** 2).sqrt()


def step_schedule(kind: str, domain, steps: int):
    u = torch.linspace(0, 1, steps + 1, dtype=torch.float32)
    if kind == "uniform":
        return torch.linspace(0, domain - 1, steps + 1, dtype=torch.long)
    if kind == "uniform_continuous":
        return torch.linspace(domain[0], domain[1], steps + 1, dtype=torch.float32)
    if kind == "quad_continuous":
        return (domain[1] - domain[0]) * (u ** 1.5) + domain[0]
    raise ValueError(kind)


# ------------------------------------------------------------------------------------------------ #
# the denoising loop                                                                                 #
# ------------------------------------------------------------------------------------------------ #
def vp_sample(forward, prior, noise: List[torch.Tensor], *, solver: str, sample_steps: int, discrete: bool,
              diffusion_steps: int = 1000, epsilon: float = 1e-3, noise_schedule: str = "cosine",
              sample_step_schedule: Optional[str] = None, temperature: float = 1.0, predict_noise: bool = True,
              fix_mask=None, x_max=None, x_min=None, cond=None, w_cfg: float = 0.0,
              diffusion_x_sampling_steps: int = 0):
    """Literal restatement of ``{Discrete,Continuous}DiffusionSDE.sample`` with recorded noise.
    ``forward(x, t, cond_or_None) -> prediction``.  Returns x0."""
    sched_fn = cosine_schedule if noise_schedule == "cosine" else linear_schedule
    draws = iter(noise)
    fm = fix_mask if fix_mask is not None else 0.
    n = prior.shape[0]
    xt = next(draws) * temperature
    xt = xt * (1. - fm) + prior * fm

    if discrete:
        alpha_all, sigma_all = sched_fn(torch.linspace(epsilon, 1.0, diffusion_steps))
        sched = step_schedule(sample_step_schedule or "uniform", diffusion_steps, sample_steps)
        alphas, sigmas = alpha_all[sched], sigma_all[sched]
        t_dtype = torch.long
    else:
        t_range = [epsilon, 0.9946] if noise_schedule == "cosine" else [epsilon, 1.]
        sched = step_schedule(sample_step_schedule or "uniform_continuous", t_range, sample_steps)
        alphas, sigmas = sched_fn(sched)
        t_dtype = torch.float32
    log_snr = torch.log(alphas / sigmas)
    hs = torch.zeros_like(log_snr)
    hs[1:] = log_snr[:-1] - log_snr[1:]
    stds = torch.zeros((sample_steps + 1,))
    stds[1:] = sigmas[:-1] / sigmas[1:] * (1 - (alphas[1:] / alphas[:-1]) ** 2).sqrt()

    buffer = []
    for i in reversed([1] * diffusion_x_sampling_steps + list(range(1, sample_steps + 1))):
        t = torch.full((n,), sched[i], dtype=t_dtype)
        a_i, s_i = alphas[i], sigmas[i]
        # classifier-free guidance (Q6)
        if w_cfg != 0.0 and w_cfg != 1.0:
            both = forward(xt.repeat(2, 1, 1), t.repeat(2), torch.cat([cond, torch.zeros_like(cond)], 0))
            pred = w_cfg * both[:n] + (1 - w_cfg) * both[n:]
        elif w_cfg == 0.0:
            pred = w_cfg * 0. + (1 - w_cfg) * forward(xt, t, None)
        else:
            pred = w_cfg * forward(xt, t, cond) + (1 - w_cfg) * 0.
        # clip
        if x_max is not None or x_min is not None:
            if predict_noise:
                upper = (xt - a_i * x_min) / s_i if x_min is not None else None
                lower = (xt - a_i * x_max) / s_i if x_max is not None else None
                pred = pred.clip(lower, upper)
            else:
                pred = pred.clip(x_min, x_max)
        eps = pred if predict_noise else (xt - a_i * pred) / s_i
        xth = pred if not predict_noise else (xt - s_i * pred) / a_i
        a_p, s_p, h = alphas[i - 1], sigmas[i - 1], hs[i]
        if solver == "ddpm":
            xt = (a_p / a_i) * (xt - s_i * eps) + (s_p ** 2 - stds[i] ** 2 + 1e-8).sqrt() * eps
            if i > 1:
                xt = xt + stds[i] * next(draws)
        elif solver == "ddim":
            xt = a_p * ((xt - s_i * eps) / a_i) + s_p * eps
        elif solver == "ode_dpmsolver_1":
            xt = (a_p / a_i) * xt - s_p * torch.expm1(h) * eps
        elif solver == "ode_dpmsolver++_1":
            xt = (s_p / s_i) * xt - a_p * torch.expm1(-h) * xth
        elif solver == "ode_dpmsolver++_2M":
            buffer.append(xth)
            if i < sample_steps:
                r = hs[i + 1] / h
                dd = (1 + 0.5 / r) * buffer[-1] - 0.5 / r * buffer[-2]
                xt = (s_p / s_i) * xt - a_p * torch.expm1(-h) * dd
            else:
                xt = (s_p / s_i) * xt - a_p * torch.expm1(-h) * xth
        elif solver == "sde_dpmsolver_1":
            xt = (a_p / a_i) * xt - 2 * s_p * torch.expm1(h) * eps + s_p * torch.expm1(2 * h).sqrt() * next(draws)
        elif solver in ("sde_dpmsolver++_1", "sde_dpmsolver++_2M"):
            v = xth
            if solver.endswith("2M"):
                buffer.append(xth)
                if i < sample_steps:
                    r = hs[i + 1] / h
                    v = (1 + 0.5 / r) * buffer[-1] - 0.5 / r * buffer[-2]
            xt = ((s_p / s_i) * (-h).exp() * xt - a_p * torch.expm1(-2 * h) * v +
                  s_p * (-torch.expm1(-2 * h)).sqrt() * next(draws))
        else:
            raise ValueError(solver)
        xt = xt * (1. - fm) + prior * fm
    if x_max is not None or x_min is not None:
        xt = xt.clip(x_min, x_max)
    return xt


# ------------------------------------------------------------------------------------------------ #
# convenience: run a golden case end to end                                                          #
# ------------------------------------------------------------------------------------------------ #
def run_case(name: str, sd=None, batch_slice=None, threads: Optional[int] = None):
    """Run oracle/cases.py case `name` through this port.  Returns x0 as numpy."""
    from . import cases
    from cleandiffuser_amd.utils.synth import synth_state_dict
    c = cases.CASES[name]
    inp = cases.make_inputs(name)
    net_kw = c["net"][1]
    if sd is None:
        lib = cases.lib_namespace("amd")                    # only to learn parameter names/shapes
        shapes = getattr(lib, c["net"][0])(**net_kw).state_dict()
        sd = synth_state_dict(shapes, 0)
    fwd = make_forward(sd, net_kw)
    kw = dict(c["sample"])
    skw = dict(c["solver"][1])
    h, d = c["horizon"], net_kw["in_dim"]
    clip = c.get("clip")
    with torch.no_grad():
        x = vp_sample(
            fwd, torch.from_numpy(inp["prior"]), [torch.from_numpy(z) for z in inp["noise"]],
            solver=kw["solver"], sample_steps=kw["sample_steps"], discrete=c["solver"][0] == "DiscreteDiffusionSDE",
            diffusion_steps=skw.get("diffusion_steps", 1000), noise_schedule=skw.get("noise_schedule", "cosine"),
            sample_step_schedule=kw.get("sample_step_schedule"), temperature=kw.get("temperature", 1.0),
            predict_noise=skw.get("predict_noise", True),
            fix_mask=torch.from_numpy(inp["fix_mask"])[None] if inp["fix_mask"] is not None else None,
            x_max=torch.full((1, h, d), float(clip)) if clip else None,
            x_min=torch.full((1, h, d), -float(clip)) if clip else None,
            cond=torch.from_numpy(inp["cond"]) if inp["cond"] is not None else None,
            w_cfg=kw.get("w_cfg", 0.0), diffusion_x_sampling_steps=kw.get("diffusion_x_sampling_steps", 0))
    return x.numpy()


def make_forward(sd, net_kw):
    n_levels = len(net_kw["dim_mult"])

    def fwd(x, t, cond):
        return janner_forward(sd, x, t, cond, emb_dim=net_kw["emb_dim"], kernel_size=net_kw["kernel_size"],
                              n_levels=n_levels)
    return fwd
