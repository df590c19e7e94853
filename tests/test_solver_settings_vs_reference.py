"""Solver settings nobody wrote a fixture for: seeded random (solver class, schedule, solver name, step count, step schedule,
temperature, prediction type, clipping, fix mask, classifier-free-guidance weight) drawn here and run THREE ways on the CPU with the same
synthetic weights and the same replayed Gaussian draws:

  1. the real reference, imported from /root/reference (build container only: the test skips where the tree is absent);
  2. this package's solver classes (the host loop the GPU path shares its step plans with);
  3. the plan's ``cdx_step`` records interpreted the way the device applies them (oracle/step_sim.py) -- what the program kernel and
     the big-batch executors run.

(1) vs (2) must agree to float rounding including the number of draws consumed; (3) must land on (1) as well.  A setting the reference
rejects must be rejected by this package with the same exception type.  A few hundred draws of this sweep ran clean when the file was
written; the suite keeps a bounded sample.
"""
import random

import numpy as np
import pytest
import torch

from oracle import cases, ref_import

pytestmark = pytest.mark.skipif(not ref_import.available(), reason="reference tree not present on this box")


def _draw(rng):
    kind = rng.choice(["DiscreteDiffusionSDE", "DiscreteDiffusionSDE", "ContinuousDiffusionSDE", "ContinuousEDM"])
    net = rng.choice([cases.JANNER_TINY, ("IDQLMlp", dict(obs_dim=0, act_dim=5, emb_dim=16, hidden_dim=32, n_blocks=1))])
    c = dict(net=net, batch=rng.randint(1, 4))
    if net[0] == "JannerUNet1d":
        c["horizon"] = 8
        if rng.random() < 0.5:
            c["fix_obs"] = 3
    else:
        c["x_shape"] = (5,)
    if rng.random() < 0.6:
        c["clip"] = rng.choice([0.5, 1.0, 3.0])
    if kind == "DiscreteDiffusionSDE":
        T = rng.choice([5, 10, 20, 50, 100])
        skw = dict(diffusion_steps=T, predict_noise=rng.random() < 0.5, noise_schedule=rng.choice(["cosine", "linear"]))
        samp = dict(solver=rng.choice(cases._ALL_SOLVERS), sample_steps=rng.randint(1, min(T, 12)),
                    sample_step_schedule=rng.choice(["uniform", "quad"]))
    elif kind == "ContinuousDiffusionSDE":
        skw = dict(predict_noise=rng.random() < 0.5, noise_schedule=rng.choice(["cosine", "linear"]))
        samp = dict(solver=rng.choice(cases._ALL_SOLVERS), sample_steps=rng.randint(1, 12),
                    sample_step_schedule=rng.choice(["uniform_continuous", "quad_continuous"]))
    else:
        skw, samp = dict(), dict(solver=rng.choice(["euler", "heun"]), sample_steps=rng.randint(1, 12))
    if rng.random() < 0.5:
        samp["temperature"] = rng.choice([0.0, 0.3, 0.7, 1.0, 1.5])
    if rng.random() < 0.3:
        samp["diffusion_x_sampling_steps"] = rng.randint(0, 3)
    if net[0] == "JannerUNet1d" and rng.random() < 0.4:
        c["cond_dim"] = 16
        samp["w_cfg"] = rng.choice([0.0, 1.0, 1.0, 2.0, 0.5])
    c["solver"], c["sample"] = (kind, skw), samp
    return c


def _sample(lib, name):
    torch.manual_seed(1234)
    agent, module = cases.build(lib, name)
    inp = cases.make_inputs(name)
    used = [0]

    def counting(noise):
        for z in noise:
            used[0] += 1
            yield z
    with cases.replay_randn(counting(inp["noise"])):
        x, _ = agent.sample(torch.from_numpy(inp["prior"]), **cases.sample_kwargs(name, inp))
    return x.detach().numpy(), used[0]


@pytest.mark.parametrize("seed", range(8))
def test_random_solver_settings_three_ways(seed, amd_lib, monkeypatch):
    from cleandiffuser_amd.engine import dispatch
    from oracle import step_sim
    ref = cases.lib_namespace("reference")
    rng = random.Random(7000 + seed)
    for i in range(5):
        name = f"_random_{seed}_{i}"
        c = cases.CASES[name] = _draw(rng)
        try:
            try:
                want, n_ref = _sample(ref, name)
            except Exception as e:                                   # the reference rejects the setting: so must this package
                with pytest.raises(type(e)):
                    _sample(amd_lib, name)
                continue
            got, n_amd = _sample(amd_lib, name)
            assert n_amd == n_ref, c
            if not np.isfinite(want).all():                           # (repeated timesteps of a quad schedule: h = 0 in the reference too)
                assert (np.isfinite(got) == np.isfinite(want)).all(), c
                continue
            scale = max(1.0, float(np.abs(want).max()))
            assert float(np.abs(got - want).max()) < 2e-5 * scale, c
            # the step records, as the device applies them
            seen = {}

            def capture(solver, model, plan, xt, prior, cond_vec, w_cfg, *a, **k):
                seen.update(plan=plan, xt=xt.clone(), cond=cond_vec, w_cfg=w_cfg)
                return None
            monkeypatch.setattr(dispatch, "try_fused_edm" if c["solver"][0] == "ContinuousEDM" else "try_fused_sample", capture)
            agent, module = cases.build(amd_lib, name)
            inp = cases.make_inputs(name)
            agent.sample(torch.from_numpy(inp["prior"]), noise=list(inp["noise"][:n_amd]), **cases.sample_kwargs(name, inp))
            monkeypatch.undo()
            assert "plan" in seen, c
            fm = torch.from_numpy(inp["fix_mask"])[None] if inp["fix_mask"] is not None else None
            x = step_sim.run_plan(seen["plan"], lambda x, t, cnd: module(x, t, cnd), seen["xt"],
                                  predict_noise=bool(getattr(agent, "predict_noise", False)), prior=torch.from_numpy(inp["prior"]),
                                  fix_mask=fm, noise=[torch.from_numpy(v) for v in inp["noise"][1:n_amd]], cond=seen["cond"],
                                  w_cfg=seen["w_cfg"], x_min=getattr(agent, "x_min", None), x_max=getattr(agent, "x_max", None))
            if getattr(agent, "clip_pred", False):
                x = x.clip(agent.x_min, agent.x_max)
            assert float(np.abs(x.numpy() - want).max()) < 1e-4 * scale, c
        finally:
            del cases.CASES[name]


@pytest.mark.parametrize("seed", range(4))
def test_random_classifier_guided_settings_match_the_reference(seed, amd_lib):
    """Classifier guidance (w_cg, CumRewClassifier over a HalfJannerUNet1d) under random solver settings: samples, draws consumed and the
    final log_p of the reference and of this package's host loop (reference diffusionsde.py:526-601)."""
    ref = cases.lib_namespace("reference")
    rng = random.Random(8000 + seed)
    for i in range(5):
        name = f"_random_cg_{seed}_{i}"
        c = _draw(rng)
        while c["solver"][0] == "ContinuousEDM" or c["net"][0] != "JannerUNet1d" or "cond_dim" in c:
            c = _draw(rng)
        c["classifier"] = dict(kernel_size=rng.choice([3, 5]))
        c["sample"]["w_cg"] = rng.choice([0.0, 0.01, 0.1, 1.0, 3.0])
        cases.CASES[name] = c
        try:
            outs = []
            for lib in (ref, amd_lib):
                torch.manual_seed(1234)
                agent, _ = cases.build(lib, name)
                inp = cases.make_inputs(name)
                used = [0]

                def counting(noise):
                    for z in noise:
                        used[0] += 1
                        yield z
                with cases.replay_randn(counting(inp["noise"])):
                    x, log = agent.sample(torch.from_numpy(inp["prior"]), **cases.sample_kwargs(name, inp))
                outs.append((x.detach().numpy(), used[0], log["log_p"].detach().numpy()))
            (want, n_ref, lp_ref), (got, n_amd, lp_amd) = outs
            assert n_ref == n_amd, c
            if not np.isfinite(want).all():
                assert (np.isfinite(got) == np.isfinite(want)).all(), c
                continue
            assert float(np.abs(got - want).max()) < 5e-5 * max(1.0, float(np.abs(want).max())), c
            assert float(np.abs(lp_amd - lp_ref).max()) < 1e-4 * max(1.0, float(np.abs(lp_ref).max())), c
        finally:
            del cases.CASES[name]
