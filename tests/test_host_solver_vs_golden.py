"""Host logic (plan compiler + PyTorch executor + module mirrors) against fixtures produced by the REAL reference
(oracle/gen_golden.py).  CPU only.  Tolerance: the mirrors evaluate the same ATen ops in the same order, so this
is held to 2e-6 abs/rel -- far inside the 1e-4 parity budget the device path is judged on."""
import numpy as np
import pytest
import torch

from oracle import cases
from conftest import golden_path


@pytest.mark.parametrize("name", list(cases.CASES))
def test_torch_executor_matches_reference_fixture(name, amd_lib):
    gold = np.load(golden_path(name))
    agent, net = cases.build(amd_lib, name)
    inp = cases.make_inputs(name)
    kw = cases.sample_kwargs(name, inp)
    n_draws = int(gold["n_draws"])
    x, log = cases.sampler_of(agent, name)(torch.from_numpy(inp["prior"]), noise=list(inp["noise"][:n_draws]), **kw)
    # autograd's conv backward sums in a thread-dependent order: guided cases get 2e-5 instead of 2e-6
    tol = 2e-5 if cases.CASES[name]["sample"].get("w_cg") else 2e-6
    np.testing.assert_allclose(x.numpy(), gold["x_out"], rtol=tol, atol=tol)
    if "log_p" in gold.files:                      # Diffuser tail: classifier score of the finished trajectories
        np.testing.assert_allclose(log["log_p"].numpy(), gold["log_p"], rtol=2e-6, atol=2e-6)
        assert int(log["log_p"].argmax()) == int(gold["log_p"].argmax())


def test_seeded_rng_path_matches_replay(amd_lib):
    """Without ``noise=`` the draws come from torch's generator in the reference's call order."""
    name = "janner_tiny_disc_ddpm"
    agent, _ = cases.build(amd_lib, name)
    inp = cases.make_inputs(name)
    kw = cases.sample_kwargs(name, inp)
    prior = torch.from_numpy(inp["prior"])
    torch.manual_seed(7)
    zs = [torch.randn_like(prior) for _ in range(5)]
    torch.manual_seed(7)
    x_rng, _ = agent.sample(prior, **kw)
    x_rep, _ = agent.sample(prior, noise=zs, **kw)
    assert torch.equal(x_rng, x_rep)


def test_short_noise_list_raises(amd_lib):
    name = "janner_tiny_disc_ddpm"
    agent, _ = cases.build(amd_lib, name)
    inp = cases.make_inputs(name)
    with pytest.raises(ValueError):
        agent.sample(torch.from_numpy(inp["prior"]), noise=list(inp["noise"][:2]), **cases.sample_kwargs(name, inp))


def test_bad_solver_and_schedule_raise(amd_lib):
    agent, _ = cases.build(amd_lib, "janner_tiny_disc_ddim")
    prior = torch.zeros(2, 8, 6)
    with pytest.raises(AssertionError):
        agent.sample(prior, solver="nope", n_samples=2)
    with pytest.raises(ValueError):
        agent.sample(prior, solver="ddim", n_samples=2, sample_step_schedule="nope")
    with pytest.raises(ValueError):
        amd_lib.DiscreteDiffusionSDE(amd_lib.JannerUNet1d(6, 16, 16, 3, [1, 2]), diffusion_steps=2000)


@pytest.mark.parametrize("predict_noise", [True, False])
def test_legacy_ddpm_plan_matches_posterior_step(predict_noise, amd_lib):
    """engine/plan.py's legacy records reproduce DDPM._posterior_step (the arithmetic the device applies)."""
    from cleandiffuser_amd.engine import plan as PL
    net = amd_lib.DQLMlp(3, 2)
    agent = amd_lib.DDPM(net, diffusion_steps=7, predict_noise=predict_noise)
    plan = PL.build_legacy_ddpm_plan(agent.beta, agent.alpha, agent.bar_alpha, predict_noise, extra_sample_steps=2)
    assert [s.t for s in plan.steps] == [6, 5, 4, 3, 2, 1, 0, 0, 0]
    assert [s.noise for s in plan.steps] == [True] * 6 + [False] * 3
    g = torch.Generator().manual_seed(0)
    x, p = torch.randn(4, 2, generator=g), torch.randn(4, 2, generator=g)
    for st in plan.steps[:7]:
        mean, std = agent._posterior_step(x, p, st.t)
        k0, k1, k2, k3, _ = st.k
        mine = k0 * (x - k1 * p) if predict_noise else k0 * (k1 * x + k2 * p)
        assert torch.allclose(mine, mean, rtol=1e-6, atol=1e-6)
        assert abs(float(std) - k3) < 1e-7


@pytest.mark.parametrize("name", ["idql_cfg5_edm_euler", "idql_cfg5_edm_heun"])
def test_edm_plan_records_reproduce_reference(name, amd_lib):
    """The cdx_step records of build_edm_plan (kinds 5/6), applied with plain torch arithmetic exactly as
    csrc/cdx_bigbatch.hip's solver_step_kernel applies them, land on the real reference's samples."""
    from cleandiffuser_amd.engine.plan import KIND_EDM_EULER, build_edm_plan
    gold = np.load(golden_path(name))
    c = cases.CASES[name]
    agent, net = cases.build(amd_lib, name)
    inp = cases.make_inputs(name)
    plan = build_edm_plan(agent.sigma_data, agent.sigma_min, agent.sigma_max, agent.rho, c["sample"]["sample_steps"],
                          c["sample"]["solver"], c["sample"].get("diffusion_x_sampling_steps", 0))
    x = torch.from_numpy(inp["noise"][0]) * agent.sigma_max * c["sample"].get("temperature", 1.0)
    slope_old = x_old = None
    with torch.no_grad():
        for st in plan.steps:
            t = torch.full((x.shape[0],), st.t, dtype=torch.float32)
            F = net(st.alpha * x, t, None)
            d = st.k[0] * x + st.k[1] * F
            s = (x - d) / st.k[2]
            if st.kind == KIND_EDM_EULER:
                if st.push:
                    slope_old, x_old = s, x
                x = x - s * st.k[3]
            else:
                x = x_old - (slope_old + s) / 2.0 * st.k[3]
    np.testing.assert_allclose(x.numpy(), gold["x_out"], rtol=1e-4, atol=1e-4)


LEGACY_PLAN_CASES = [n for n, c in cases.CASES.items() if c["solver"][0] in ("DPMSolver", "EDM")]


@pytest.mark.parametrize("name", LEGACY_PLAN_CASES)
def test_legacy_plan_records_reproduce_reference(name, amd_lib):
    """build_legacy_dpmsolver_plan / build_legacy_edm_plan, interpreted with the device's step semantics (oracle/step_sim.py),
    land on the samples of the real reference's DPMSolver / EDM classes (tests/golden)."""
    from cleandiffuser_amd.engine import plan as P
    from oracle import step_sim
    gold = np.load(golden_path(name))
    c = cases.CASES[name]
    agent, net = cases.build(amd_lib, name)
    inp = cases.make_inputs(name)
    s = c["sample"]
    prior = torch.from_numpy(inp["prior"])
    fm = torch.from_numpy(inp["fix_mask"])[None] if inp["fix_mask"] is not None else None
    z = [torch.from_numpy(v) for v in inp["noise"]]
    cond = None
    if inp["cond"] is not None:
        with torch.no_grad():
            cond = agent.model_ema["condition"](torch.from_numpy(inp["cond"]), None)
    if c["solver"][0] == "DPMSolver":
        S, kappa = s["sample_steps"], s.get("kappa", 1.0)
        idx = torch.arange(S + 1)
        t = ((S - idx) / S * agent.t_range[1] ** (1 / kappa) + idx / S * agent.t_range[0] ** (1 / kappa)) ** kappa
        alphas = agent.alpha_schedule(t)
        plan = P.build_legacy_dpmsolver_plan(t, alphas, (1 - alphas ** 2).sqrt(), s["sampler"], S, s.get("extra_sample_steps", 0))
        x0 = z[0] * s.get("temperature", 1.0)
        kw = dict(predict_noise=agent.predict_noise, x_min=agent.x_min, x_max=agent.x_max)
    else:
        agent.set_sample_steps(s["sample_steps"])
        plan = P.build_legacy_edm_plan(agent.sigma_data, agent.sigma_s, s["solver"], s.get("extra_sample_steps", 0))
        x0 = z[0] * agent.sigma_s[0]
        kw = dict(predict_noise=False)
    if fm is not None:
        x0 = x0 * (1 - fm) + prior * fm
    x = step_sim.run_plan(plan, net, x0, prior=prior, fix_mask=fm, noise=z[1:], cond=cond, w_cfg=s.get("w_cfg", 0.0), **kw)
    if c.get("clip") and c["solver"][0] == "DPMSolver":
        x = x.clip(agent.x_min, agent.x_max)
    np.testing.assert_allclose(x.numpy(), gold["x_out"], rtol=1e-4, atol=1e-4)


FLOW_CM_CASES = [n for n, c in cases.CASES.items()
                 if c["solver"][0] in ("DiscreteRectifiedFlow", "ContinuousRectifiedFlow", "ContinuousConsistencyModel")]


@pytest.mark.parametrize("name", FLOW_CM_CASES)
def test_flow_and_consistency_plans_reproduce_reference(name, amd_lib, monkeypatch):
    """The records the rectified-flow / consistency classes hand to the device (captured from the class itself by
    intercepting the dispatch hook), interpreted by oracle/step_sim.py, land on the real reference's samples."""
    from cleandiffuser_amd.engine import dispatch
    from oracle import step_sim
    gold = np.load(golden_path(name))
    c = cases.CASES[name]
    agent, net = cases.build(amd_lib, name)
    inp = cases.make_inputs(name)
    seen = {}

    def capture(solver, model, plan, xt, prior, cond_vec, w_cfg, w_cg, requires_grad, feed):
        seen.update(plan=plan, xt=xt.clone(), cond=cond_vec, w_cfg=w_cfg)
        return None                                        # -> the class continues on its PyTorch loop
    monkeypatch.setattr(dispatch, "try_fused_sample", capture)
    monkeypatch.setattr(dispatch, "try_fused_edm", capture)
    kw = cases.sample_kwargs(name, inp)
    n_draws = int(gold["n_draws"])
    x_ref, _ = agent.sample(torch.from_numpy(inp["prior"]), noise=list(inp["noise"][:n_draws]), **kw)
    plan = seen["plan"]
    fm = torch.from_numpy(inp["fix_mask"])[None] if inp["fix_mask"] is not None else None
    clip_d = c["solver"][0] == "ContinuousConsistencyModel"    # f() clips inside every evaluation; flows clip once at the end
    x = step_sim.run_plan(plan, net, seen["xt"], predict_noise=bool(plan.network_predicts_noise),
                          prior=torch.from_numpy(inp["prior"]), fix_mask=fm,
                          noise=[torch.from_numpy(v) for v in inp["noise"][1:]], cond=seen["cond"], w_cfg=seen["w_cfg"],
                          x_min=agent.x_min if clip_d else None, x_max=agent.x_max if clip_d else None)
    if agent.clip_pred and not clip_d:
        x = x.clip(agent.x_min, agent.x_max)
    np.testing.assert_allclose(x.numpy(), gold["x_out"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(x_ref.numpy(), gold["x_out"], rtol=2e-6, atol=2e-6)


def test_step_plans_are_memoised_per_settings(amd_lib):
    """A control loop calling sample() with the same settings re-derives nothing on the host: the plan object is reused, a change
    of solver / step count / Diffusion-X tail gets its own plan, and results stay equal to the reference fixture throughout."""
    name = "janner_tiny_disc_ddim"
    gold = np.load(golden_path(name))
    agent, _ = cases.build(amd_lib, name)
    inp = cases.make_inputs(name)
    kw = cases.sample_kwargs(name, inp)
    prior, zs = torch.from_numpy(inp["prior"]), list(inp["noise"][:int(gold["n_draws"])])
    x1, _ = agent.sample(prior, noise=zs, **kw)
    (plan1,) = agent._plans.values()
    x2, _ = agent.sample(prior, noise=zs, **kw)
    assert list(agent._plans.values()) == [plan1] and next(iter(agent._plans.values())) is plan1
    assert torch.equal(x1, x2)
    np.testing.assert_allclose(x1.numpy(), gold["x_out"], rtol=2e-6, atol=2e-6)
    kw2 = dict(kw, sample_steps=kw["sample_steps"] - 1)
    agent.sample(prior, noise=list(inp["noise"]), **kw2)
    agent.sample(prior, noise=list(inp["noise"]), **dict(kw, solver="ddpm"))
    agent.sample(prior, noise=list(inp["noise"]), **dict(kw, diffusion_x_sampling_steps=1))
    assert len(agent._plans) == 4
    x3, _ = agent.sample(prior, noise=zs, **kw)                       # the first plan is still the one used for the first settings
    assert torch.equal(x1, x3)
    for i in range(20):                                                # bounded: the oldest entries are dropped
        agent.sample(prior, noise=list(inp["noise"]), **dict(kw, sample_steps=2, diffusion_x_sampling_steps=i))
    assert len(agent._plans) <= 16
