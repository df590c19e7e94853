"""loss() / update() scenarios -- TEST INFRASTRUCTURE (oracle/__init__.py).

north_star names ``DiffusionModel.loss()`` as part of the surface; round 2 left it pinned by nothing but a finiteness check.
Each scenario builds one solver class (reference diffusion/diffusionsde.py:94-141, 387-397, 725-739; newedm.py:152-190;
ddpm.py:80-112) in TRAIN mode on synthetic weights, then under ``torch.manual_seed`` records

* ``loss``        -- the value of one ``loss(x0, condition)`` call (draw order: timestep, noise, label-dropout mask),
* ``upd_loss``    -- the ``loss`` entries of three consecutive ``update()`` calls (AdamW step + EMA each),
* ``grad_norm``   -- the clipped-gradient norms ``update()`` reports,
* ``p_head`` / ``ema_head`` / ``p_abs`` / ``ema_abs`` -- the first 16 entries of the first parameter and the sum of |p| over all
  parameters of ``model`` and ``model_ema`` after the three updates.

``python -m oracle.gen_train_golden`` runs them on the REAL reference (build container) and commits ``tests/golden/train_*.npz``;
the tests run the same functions on this repo's classes (CPU, and the ROCm device with the CPU generator's draws replayed through
``cpu_rng`` so both sides see identical timesteps / noise / masks).
"""
import contextlib
from typing import Callable, Dict

import numpy as np
import torch

from cleandiffuser_amd.utils import load_synth
from . import cases


@contextlib.contextmanager
def cpu_rng(device):
    """Route the random draws of loss() through the CPU generator and move them to `device`: a ROCm device has its own Philox
    stream, so seeded draws made there can never equal the reference's CPU ones."""
    if str(device) == "cpu":
        yield
        return
    orig = {k: getattr(torch, k) for k in ("randn_like", "randn", "rand", "randint")}

    def randn_like(ref, *a, **k):
        return orig["randn"](ref.shape, dtype=ref.dtype).to(ref.device)

    def _strip(fn):
        def g(*a, **k):
            dev = k.pop("device", None)
            out = fn(*a, **k)
            return out.to(dev) if dev is not None else out
        return g
    torch.randn_like = randn_like
    torch.randn, torch.rand, torch.randint = _strip(orig["randn"]), _strip(orig["rand"]), _strip(orig["randint"])
    try:
        yield
    finally:
        for k, v in orig.items():
            setattr(torch, k, v)


def _record(agent, x0, cond, device, n_updates=3):
    x0 = x0.to(device)
    cond = None if cond is None else cond.to(device)
    agent.train()
    out = {}
    with cpu_rng(device):
        torch.manual_seed(4321)
        out["loss"] = agent.loss(x0, cond).detach().reshape(1)
        losses, norms = [], []
        for _ in range(n_updates):
            log = agent.update(x0, cond)
            losses.append(float(log["loss"]))
            gn = log.get("grad_norm")
            norms.append(float(gn) if gn is not None else 0.0)
    out["upd_loss"] = torch.tensor(losses)
    out["grad_norm"] = torch.tensor(norms)
    for tag, mod in (("p", agent.model), ("ema", agent.model_ema)):
        ps = list(mod.parameters())
        out[tag + "_head"] = ps[0].detach().reshape(-1)[:16].clone()
        out[tag + "_abs"] = torch.stack([p.detach().abs().sum() for p in ps]).sum().reshape(1)
    return out


def _janner(lib, device, cls, **kw):
    net = load_synth(lib.JannerUNet1d(6, model_dim=16, emb_dim=16, dim_mult=[1, 2], kernel_size=5), 61)
    fm = torch.zeros(8, 6)
    fm[0, :4] = 1.0
    return getattr(lib, cls)(net, None, fix_mask=fm, grad_clip_norm=1.0, device=device, **kw)


def discrete(predict_noise: bool):
    def run(lib, kind, device):
        agent = _janner(lib, device, "DiscreteDiffusionSDE", diffusion_steps=50, predict_noise=predict_noise,
                        loss_weight=torch.linspace(0.5, 1.5, 6).expand(8, 6).contiguous())
        g = torch.Generator().manual_seed(1)
        return _record(agent, torch.randn(5, 8, 6, generator=g), None, device)
    return run


def continuous():
    def run(lib, kind, device):
        agent = _janner(lib, device, "ContinuousDiffusionSDE", predict_noise=True, noise_schedule="linear")
        g = torch.Generator().manual_seed(2)
        return _record(agent, torch.randn(5, 8, 6, generator=g), None, device)
    return run


def edm_conditional(dropout: float):
    """ContinuousEDM over a conditional MLP denoiser; IdentityCondition with label dropout 0.25 in train mode (the Bernoulli mask is a
    third seeded draw, reference nn_condition/base_nn_condition.py:7-12).  `dropout` = the backbone's own nn.Dropout rate: 0.1 (the
    default) consumes the CPU generator inside ATen, which a ROCm device cannot replay -- that variant is CPU-only; 0.0 runs on both."""
    def run(lib, kind, device):
        net = load_synth(lib.IDQLMlp(11, 3, emb_dim=16, hidden_dim=64, n_blocks=2, dropout=dropout), 62)
        agent = lib.ContinuousEDM(net, lib.IdentityCondition(dropout=0.25), grad_clip_norm=0.5, device=device)
        g = torch.Generator().manual_seed(3)
        return _record(agent, torch.randn(9, 3, generator=g), torch.randn(9, 11, generator=g), device)
    return run


def legacy_ddpm():
    def run(lib, kind, device):
        net = load_synth(lib.PearceMlp(6, To=1, emb_dim=32, hidden_dim=64), 63)
        cond = load_synth(lib.PearceObsCondition(11, 32, flatten=True, dropout=0.0), 64)
        agent = lib.DDPM(net, cond, diffusion_steps=20, predict_noise=False, device=device)
        g = torch.Generator().manual_seed(4)
        return _record(agent, torch.randn(7, 6, generator=g).clamp(-1, 1), torch.randn(7, 1, 11, generator=g), device)
    return run


def weighted_regression():
    """``weighted_regression_tensor`` kwarg of the new-style loss (reference diffusionsde.py:107-110; the AWR pipelines pass it)."""
    def run(lib, kind, device):
        net = load_synth(lib.DQLMlp(11, 6, emb_dim=16), 65)
        agent = lib.DiscreteDiffusionSDE(net, lib.IdentityCondition(dropout=0.0), diffusion_steps=10, predict_noise=True, device=device)
        g = torch.Generator().manual_seed(5)
        x0, c, w = torch.randn(6, 6, generator=g), torch.randn(6, 11, generator=g), torch.rand(6, generator=g)
        agent.train()
        with cpu_rng(device):
            torch.manual_seed(99)
            loss = agent.loss(x0.to(device), c.to(device), weighted_regression_tensor=w.to(device))
            log = agent.update(x0.to(device), c.to(device), weighted_regression_tensor=w.to(device))
        return {"loss": loss.detach().reshape(1), "upd_loss": torch.tensor([float(log["loss"])])}
    return run


def chiunet(model_dim: int, batch: int):
    """ChiUNet1d with a global condition under the legacy DDPM class -- the dp_pusht training step, BASELINE config 3 (reference
    pipelines/dp_pusht.py, nn_diffusion/chiunet.py:152-192, ddpm.py:80-112).  model_dim 32: GroupNorm groups of 4 / 8 / 16 channels;
    model_dim 256: the exact config-3 net (68.9 M parameters, groups of 32 / 64 / 128 channels, K up to 10 240)."""
    def run(lib, kind, device):
        net = load_synth(lib.ChiUNet1d(2, 5, 2, model_dim=model_dim, emb_dim=model_dim, dim_mult=[1, 2, 2], obs_as_global_cond=True), 66)
        agent = lib.DDPM(net, lib.IdentityCondition(dropout=0.0), diffusion_steps=20, predict_noise=True, grad_clip_norm=1.0, device=device)
        g = torch.Generator().manual_seed(6)
        return _record(agent, torch.randn(batch, 16, 2, generator=g).clamp(-1, 1), torch.randn(batch, 2, 5, generator=g), device)
    return run


def dit(d_model: int, heads: int, tokens: int, batch: int):
    """DiT1d under ContinuousDiffusionSDE with an MLP condition and label dropout -- the Decision-Diffuser training step, BASELINE
    config 4 (reference pipelines/dd_d4rl_*.py, nn_diffusion/dit.py:10-130, diffusionsde.py:94-141).  d 64: 4 heads x 16; d 320 / 10
    heads / 64 tokens: the exact config-4 net."""
    def run(lib, kind, device):
        net = load_synth(lib.DiT1d(7, emb_dim=32, d_model=d_model, n_heads=heads, depth=2, timestep_emb_type="fourier"), 67)
        cond = load_synth(lib.MLPCondition(1, 32, [32], torch.nn.SiLU(), dropout=0.25), 68)
        fm = torch.zeros(tokens, 7)
        fm[0] = 1.0
        agent = lib.ContinuousDiffusionSDE(net, cond, fix_mask=fm, predict_noise=True, noise_schedule="linear", grad_clip_norm=1.0, device=device)
        g = torch.Generator().manual_seed(7)
        return _record(agent, torch.randn(batch, tokens, 7, generator=g), torch.rand(batch, 1, generator=g), device)
    return run


def chitf(d_model: int, heads: int, layers: int, cond_layers: int, ta: int, to: int, batch: int):
    """ChiTransformer under the legacy DDPM class with the observations passed through -- the dp_* training step with the transformer
    denoiser (reference pipelines/dp_pusht.py:173-196, nn_diffusion/chitransformer.py:60-158, ddpm.py:80-112).  p_drop_attn = 0 here:
    the pipelines train with 0.3, whose draws happen inside ATen and cannot be replayed across devices (the dropout path is checked
    against autograd with shared masks in tests/test_gpu_parity.py).  d 256 / 4 heads / 8 layers / Ta 10 / To 2: the dp_pusht net;
    the small one has a TransformerEncoder over the memory tokens (n_cond_layers 2)."""
    def run(lib, kind, device):
        net = load_synth(lib.ChiTransformer(3, 5, ta, to, d_model=d_model, nhead=heads, num_layers=layers, p_drop_attn=0.0,
                                            n_cond_layers=cond_layers), 69)
        agent = lib.DDPM(net, lib.IdentityCondition(dropout=0.0), diffusion_steps=20, predict_noise=True, grad_clip_norm=1.0, device=device)
        g = torch.Generator().manual_seed(8)
        return _record(agent, torch.randn(batch, ta, 3, generator=g).clamp(-1, 1), torch.randn(batch, to, 5, generator=g), device)
    return run


def sfbc():
    """SfBCUNet under ContinuousDiffusionSDE with the state as condition -- the behaviour-policy training step of SfBC (reference
    pipelines/sfbc_d4rl_mujoco.py:61-70, nn_diffusion/sfbc_unet.py:9-82, diffusionsde.py:94-141)."""
    def run(lib, kind, device):
        net = load_synth(lib.SfBCUNet(4, emb_dim=32, hidden_dims=[64, 32, 16]), 70)
        cond = load_synth(lib.MLPCondition(9, 32, [32], torch.nn.SiLU(), dropout=0.0), 71)
        agent = lib.ContinuousDiffusionSDE(net, cond, predict_noise=True, noise_schedule="linear", grad_clip_norm=1.0, device=device)
        g = torch.Generator().manual_seed(9)
        return _record(agent, torch.randn(8, 4, generator=g), torch.randn(8, 9, generator=g), device)
    return run


def classifier_cumrew(horizon: int, model_dim: int, batch: int, lr: float = 3e-3):
    """``CumRewClassifier(HalfJannerUNet1d).update`` -- the half of every Diffuser training iteration next to ``update()`` (reference
    pipelines/diffuser_d4rl_mujoco.py:88-91 -> diffusionsde.py:143-149 -> classifier/base.py:47-58, rew_classifiers.py:16-24): MSE on
    the return target, torch.optim.Adam (lr 2e-4, L2 weight decay 1e-4), EMA 0.995.  Records as the solver scenarios do: one loss
    value, three updates, parameter / EMA checksums (``grad_norm``: zeros -- the class clips nothing)."""
    def run(lib, kind, device):
        net = load_synth(lib.HalfJannerUNet1d(horizon, 6, out_dim=1, kernel_size=3, model_dim=model_dim, emb_dim=model_dim,
                                              dim_mult=(1, 2, 2)), 71)
        clf = lib.CumRewClassifier(net, device=device, optim_params={"lr": lr, "weight_decay": 1e-2})     # (steps large enough to resolve)
        g = torch.Generator().manual_seed(5)
        x = torch.randn(batch, horizon, 6, generator=g).to(device)
        t = torch.randint(0, 50, (batch,), generator=g).to(device)
        r = torch.randn(batch, 1, generator=g).to(device)
        clf.train()
        out = {"loss": clf.loss(x, t, r).detach().reshape(1)}
        out["upd_loss"] = torch.tensor([float(clf.update(x, t, r)["loss"]) for _ in range(3)])
        out["grad_norm"] = torch.zeros(3)
        for tag, mod in (("p", clf.model), ("ema", clf.model_ema)):
            ps = list(mod.parameters())
            out[tag + "_head"] = ps[0].detach().reshape(-1)[:16].clone()
            out[tag + "_abs"] = torch.stack([p.detach().abs().sum() for p in ps]).sum().reshape(1)
        return out
    return run


HEAVY = {"chiunet_cfg3", "dit_cfg4"}          # minutes of CPU work: fixture from the real reference, checked on the device only

SCENARIOS: Dict[str, Callable] = {
    "chiunet_ddpm": chiunet(32, 5), "chiunet_cfg3": chiunet(256, 4), "dit_small": dit(64, 4, 16, 5), "dit_cfg4": dit(320, 10, 64, 4),
    "chitf_small": chitf(64, 4, 2, 2, 6, 3, 5), "chitf_pusht": chitf(256, 4, 8, 0, 10, 2, 6), "sfbc_continuous": sfbc(),
    "discrete_eps": discrete(True), "discrete_x0": discrete(False), "continuous_eps": continuous(),
    "classifier_cumrew": classifier_cumrew(16, 16, 6), "classifier_cfg2": classifier_cumrew(32, 32, 8, lr=5e-4),
    "edm_conditional": edm_conditional(0.1), "edm_conditional_nodrop": edm_conditional(0.0), "legacy_ddpm": legacy_ddpm(), "weighted_regression": weighted_regression(),
}


def run(name: str, lib_kind: str, device="cpu"):
    torch.manual_seed(1234)
    out = SCENARIOS[name](cases.lib_namespace(lib_kind), lib_kind, device)
    return {k: v.detach().cpu().numpy().astype(np.float32) for k, v in out.items()}
