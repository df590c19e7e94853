"""loss() / update() settings nobody wrote a fixture for: seeded random (solver class, denoiser, prediction type, noise schedule, fix mask,
loss weight, gradient clipping, EMA rate, optimiser parameters, batch) drawn here and run on the CPU through the REAL reference
(imported from /root/reference; the test skips where the tree is absent) and through this package with the same synthetic weights and
the same seeded timestep / noise / label-dropout draws (oracle/train_cases.py:_record): the loss value, three AdamW + EMA updates, the
clipped-gradient norms and the parameter / EMA checksums must agree to float rounding.  Every denoiser the reference's pipelines
train is in the draw (reference diffusion/diffusionsde.py:94-141, newedm.py:152-190, ddpm.py:80-112, rectifiedflow.py)."""
import random

import numpy as np
import pytest
import torch

from cleandiffuser_amd.utils import load_synth
from oracle import cases, ref_import, train_cases

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference tree not present on this box")


def _denoiser(rng):
    """-> (constructor(lib), x0 shape per sample, condition shape per sample | None)"""
    pick = rng.choice(["janner", "janner_cond", "dql", "idql", "dit", "chiunet", "chitf", "pearce", "sfbc", "dvinv"])
    if pick == "janner":
        return pick, lambda lib: lib.JannerUNet1d(6, model_dim=16, emb_dim=16, dim_mult=[1, 2], kernel_size=rng_k[0]), (8, 6), None
    if pick == "janner_cond":
        return pick, lambda lib: lib.JannerUNet1d(5, model_dim=16, emb_dim=16, dim_mult=[1, 2, 2], kernel_size=5), (8, 5), (16,)
    if pick == "dql":
        return pick, lambda lib: lib.DQLMlp(11, 6, emb_dim=16), (6,), (11,)
    if pick == "dvinv":
        return pick, lambda lib: lib.DVInvMlp(7, 3, emb_dim=16, hidden_dim=32), (3,), (14,)
    if pick == "idql":
        return pick, lambda lib: lib.IDQLMlp(9, 4, emb_dim=16, hidden_dim=32, n_blocks=2, dropout=0.0), (4,), (9,)
    if pick == "dit":
        return pick, lambda lib: lib.DiT1d(7, emb_dim=32, d_model=32, n_heads=2, depth=1, timestep_emb_type="fourier"), (8, 7), (32,)
    if pick == "chiunet":
        return pick, lambda lib: lib.ChiUNet1d(2, 5, 2, model_dim=16, emb_dim=16, dim_mult=[1, 2], obs_as_global_cond=True), (8, 2), (2, 5)
    if pick == "chitf":
        return pick, lambda lib: lib.ChiTransformer(3, 5, 6, 2, d_model=32, nhead=2, num_layers=1, p_drop_attn=0.0), (6, 3), (2, 5)
    if pick == "pearce":
        return pick, lambda lib: lib.PearceMlp(4, To=2, emb_dim=16, hidden_dim=32), (4,), (2, 16)
    return pick, lambda lib: lib.SfBCUNet(5, emb_dim=16, hidden_dims=[32, 16]), (5,), (16,)


rng_k = [5]


def _draw(rng):
    rng_k[0] = rng.choice([3, 5])
    name, make, xs, cs = _denoiser(rng)
    kind = rng.choice(["DiscreteDiffusionSDE", "ContinuousDiffusionSDE", "ContinuousEDM", "DDPM", "ContinuousRectifiedFlow", "DiscreteRectifiedFlow"])
    kw = {}
    if kind in ("DiscreteDiffusionSDE", "DDPM", "DiscreteRectifiedFlow"):
        kw["diffusion_steps"] = rng.choice([5, 20, 100])
    if kind in ("DiscreteDiffusionSDE", "ContinuousDiffusionSDE"):
        kw.update(noise_schedule=rng.choice(["cosine", "linear"]), predict_noise=rng.random() < 0.5)
    if kind == "DDPM":
        kw["predict_noise"] = rng.random() < 0.5
    if rng.random() < 0.6:
        kw["grad_clip_norm"] = rng.choice([0.1, 1.0, 10.0])
    if rng.random() < 0.5:
        kw["ema_rate"] = rng.choice([0.9, 0.99, 0.999])
    if rng.random() < 0.4:
        kw["optim_params"] = dict(lr=rng.choice([1e-4, 1e-3]), weight_decay=rng.choice([0.0, 1e-2]))
    mask = loss_w = None
    if len(xs) == 2 and rng.random() < 0.5:
        mask = torch.zeros(xs)
        mask[0, :xs[1] // 2] = 1.0
    if rng.random() < 0.4:
        loss_w = torch.linspace(0.5, 1.5, xs[-1]).expand(xs).contiguous()
    batch = rng.randint(2, 6)
    return dict(name=name, make=make, kind=kind, kw=kw, mask=mask, loss_w=loss_w, xs=xs, cs=cs, batch=batch, seed=rng.randint(0, 10 ** 6),
                label_dropout=rng.choice([0.0, 0.25]) if cs is not None else 0.0)


def _run(lib, d):
    torch.manual_seed(99)
    net = load_synth(d["make"](lib), d["seed"] % 1000)
    cond_net = lib.IdentityCondition(dropout=d["label_dropout"]) if d["cs"] is not None else None
    kw = dict(d["kw"])
    if d["mask"] is not None:
        kw["fix_mask"] = d["mask"].clone()
    if d["loss_w"] is not None:
        kw["loss_weight"] = d["loss_w"].clone()
    agent = getattr(lib, d["kind"])(net, cond_net, device="cpu", **kw)
    g = torch.Generator().manual_seed(d["seed"])
    x0 = torch.randn(d["batch"], *d["xs"], generator=g).clamp(-2, 2)
    cond = torch.randn(d["batch"], *d["cs"], generator=g) if d["cs"] is not None else None
    out = train_cases._record(agent, x0, cond, "cpu")
    return {k: v.detach().numpy().astype(np.float64) for k, v in out.items()}


@pytest.mark.parametrize("seed", range(6))
def test_random_training_settings_against_the_imported_reference(seed, amd_lib):
    ref = cases.lib_namespace("reference")
    rng = random.Random(9100 + seed)
    seen = []
    for _ in range(5):
        d = _draw(rng)
        tag = f"{d['kind']} / {d['name']} / {d['kw']} / mask {d['mask'] is not None} / weight {d['loss_w'] is not None} / B {d['batch']}"
        try:
            want = _run(ref, d)
        except Exception as e:  # noqa: BLE001 -- a setting the reference rejects must be rejected here with the same exception type
            with pytest.raises(type(e)):
                _run(amd_lib, d)
            seen.append((tag, "rejected: " + type(e).__name__))
            continue
        got = _run(amd_lib, d)
        assert set(got) == set(want), tag
        for k in want:
            np.testing.assert_allclose(got[k], want[k], rtol=5e-6, atol=5e-6, err_msg=f"{tag}: {k}")
        seen.append((tag, "ok"))
    assert seen
