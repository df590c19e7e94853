"""Golden-case specifications shared by the fixture generator (run on the REAL reference) and the tests
(run on cleandiffuser_amd and on the oracle ports).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

A case is pure data; ``build(lib, case)`` instantiates it with whichever library namespace is passed
(the reference's modules or this repo's mirrors), ``make_inputs(case)`` derives every input tensor from the
deterministic PCG64 streams in ``cleandiffuser_amd.utils.synth`` (numpy only, stable across machines).
"""
import contextlib
from types import SimpleNamespace

import numpy as np
import torch

from cleandiffuser_amd.utils.synth import synth_array, synth_state_dict

JANNER_CFG2 = ("JannerUNet1d", dict(in_dim=23, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5))
JANNER_TINY = ("JannerUNet1d", dict(in_dim=6, model_dim=16, emb_dim=16, dim_mult=[1, 2], kernel_size=3))
JANNER_H4 = ("JannerUNet1d", dict(in_dim=23, model_dim=32, emb_dim=32, dim_mult=[1, 4, 2], kernel_size=5))

_ALL_SOLVERS = ["ddpm", "ddim", "ode_dpmsolver_1", "ode_dpmsolver++_1", "ode_dpmsolver++_2M",
                "sde_dpmsolver_1", "sde_dpmsolver++_1", "sde_dpmsolver++_2M"]

CASES = {
    # BASELINE config 2 (north star) at a fixture-sized batch: x-prediction, fix-mask on obs of step 0, 20/20 DDIM
    "janner_cfg2_ddim": dict(
        net=JANNER_CFG2, horizon=32, batch=4, fix_obs=17,
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=20, predict_noise=False)),
        sample=dict(solver="ddim", sample_steps=20, temperature=0.5)),
    # same network, stochastic DDPM with eps-prediction and prediction clipping, S < T
    "janner_cfg2_ddpm_clip": dict(
        net=JANNER_CFG2, horizon=32, batch=3, fix_obs=17, clip=2.0,
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=20, predict_noise=True)),
        sample=dict(solver="ddpm", sample_steps=10, temperature=1.0)),
    # shipped halfcheetah Diffuser shape: horizon 4, dim_mult [1,4,2] (C*L is not constant across levels)
    "janner_h4_ddpm": dict(
        net=JANNER_H4, horizon=4, batch=5, fix_obs=17,
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=20, predict_noise=False)),
        sample=dict(solver="ddpm", sample_steps=20, temperature=0.5)),
    # the real Diffuser tail: a CumRewClassifier(HalfJannerUNet1d) scores the finished trajectories (log_p) and the caller
    # picks arg-max candidates (reference diffusionsde.py:597-601, pipelines/diffuser_d4rl_mujoco.py:144-147)
    "janner_cfg2_diffuser_logp": dict(
        net=JANNER_CFG2, horizon=32, batch=8, fix_obs=17, classifier=dict(kernel_size=3),
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=20, predict_noise=False)),
        sample=dict(solver="ddpm", sample_steps=20, temperature=0.5)),
    # conditional backbone input (b, emb_dim) through IdentityCondition: w_cfg = 1 (one forward) ...
    "janner_tiny_cond_w1": dict(
        net=JANNER_TINY, horizon=8, batch=3, cond_dim=16, clip=3.0,
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=50, predict_noise=True)),
        sample=dict(solver="ddim", sample_steps=5, w_cfg=1.0)),
    # ... and w_cfg = 2 (reference doubles the batch: cond | zeros)
    "janner_tiny_cond_w2": dict(
        net=JANNER_TINY, horizon=8, batch=3, cond_dim=16, clip=3.0,
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=50, predict_noise=True)),
        sample=dict(solver="ode_dpmsolver++_2M", sample_steps=5, w_cfg=2.0)),
    # continuous-time solver: float timesteps -> full positional embedding, linear schedule, quad step schedule
    "janner_tiny_cont_quad": dict(
        net=JANNER_TINY, horizon=8, batch=3, clip=3.0,
        solver=("ContinuousDiffusionSDE", dict(predict_noise=True, noise_schedule="linear")),
        sample=dict(solver="sde_dpmsolver++_2M", sample_steps=6, sample_step_schedule="quad_continuous",
                    temperature=0.7, diffusion_x_sampling_steps=2)),
}
# long horizon (64 positions = two 32-column passes of the 16x16 K loop) and a single trajectory (grid of one workgroup)
CASES["janner_h64_single"] = dict(
    net=("JannerUNet1d", dict(in_dim=5, model_dim=16, emb_dim=16, dim_mult=[1, 2, 2], kernel_size=5)),
    horizon=64, batch=1, fix_obs=3,
    solver=("DiscreteDiffusionSDE", dict(diffusion_steps=10, predict_noise=False)),
    sample=dict(solver="ddim", sample_steps=4, temperature=0.9))

# ---- the other BASELINE configs at fixture size (PyTorch executor today; fused paths are later rows) ----
CASES.update({
    # config 1: PearceMlp DBC, DDPM 100 -> 20 steps here, x-prediction with clip, w_cfg = 1, PearceObsCondition
    "pearce_cfg1_ddpm": dict(
        net=("PearceMlp", dict(act_dim=6, To=1, emb_dim=64, hidden_dim=256)), x_shape=(6,), batch=5, clip=1.0,
        cond=("PearceObsCondition", dict(obs_dim=17, emb_dim=64, flatten=True, dropout=0.0), (1, 17)),
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=20, predict_noise=False)),
        sample=dict(solver="ddpm", sample_steps=20, temperature=0.5, w_cfg=1.0)),
    # config 3: ChiUNet1d Diffusion Policy through the legacy DDPM class (narrow channels, 10 steps)
    "chiunet_cfg3_legacy_ddpm": dict(
        net=("ChiUNet1d", dict(act_dim=2, obs_dim=20, To=2, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2])),
        x_shape=(16, 2), batch=3, clip=1.0, cond=("IdentityCondition", dict(dropout=0.0), (2, 20)),
        solver=("DDPM", dict(diffusion_steps=10)), legacy=True,
        sample=dict(sample_steps=10, w_cfg=1.0)),
    # config 4: DiT1d Decision Diffuser, CFG w = 2, DPM-Solver++ 2M, continuous time, fix-mask on the first token
    "dit_cfg4_cfg2_dpmpp2m": dict(
        net=("DiT1d", dict(in_dim=29, emb_dim=32, d_model=64, n_heads=4, depth=2, timestep_emb_type="fourier")),
        x_shape=(16, 29), batch=3, fix_first_token=True, clip=3.0,
        cond=("MLPCondition", dict(in_dim=1, out_dim=32, hidden_dims=[32], act="SiLU", dropout=0.25), (1,)),
        solver=("ContinuousDiffusionSDE", dict(predict_noise=True, noise_schedule="linear")),
        sample=dict(solver="ode_dpmsolver++_2M", sample_steps=10, w_cfg=2.0, temperature=0.5)),
    # config 5: IDQLMlp (SynthER ResidualMLP) through ContinuousEDM, Euler and Heun
    "idql_cfg5_edm_euler": dict(
        net=("IDQLMlp", dict(obs_dim=0, act_dim=15, emb_dim=32, hidden_dim=64, n_blocks=2)), x_shape=(15,), batch=6,
        solver=("ContinuousEDM", dict()), sample=dict(solver="euler", sample_steps=16)),
    "idql_cfg5_edm_heun": dict(
        net=("IDQLMlp", dict(obs_dim=0, act_dim=15, emb_dim=32, hidden_dim=64, n_blocks=2)), x_shape=(15,), batch=6,
        solver=("ContinuousEDM", dict()), sample=dict(solver="heun", sample_steps=8, diffusion_x_sampling_steps=1)),
    # DQL / ChiTransformer backbones under the new-style discrete solver
    "dqlmlp_ddpm": dict(
        net=("DQLMlp", dict(obs_dim=17, act_dim=6)), x_shape=(6,), batch=4, clip=1.0,
        cond=("IdentityCondition", dict(dropout=0.0), (17,)),
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=5, predict_noise=True)),
        sample=dict(solver="ddpm", sample_steps=5, w_cfg=1.0)),
    "chitransformer_ddim": dict(
        net=("ChiTransformer", dict(act_dim=2, obs_dim=20, Ta=16, To=2, d_model=64, nhead=4, num_layers=2)),
        x_shape=(16, 2), batch=3, clip=1.0, cond=("IdentityCondition", dict(dropout=0.0), (2, 20)),
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=50, predict_noise=True)),
        sample=dict(solver="ddim", sample_steps=5, w_cfg=1.0)),
})

# ---- big-batch executors (DiT1d, IDQLMlp/NewIDQLMlp): more solver/guidance combinations + the full-size networks ----
DIT_SMALL = ("DiT1d", dict(in_dim=7, emb_dim=32, d_model=64, n_heads=4, depth=2, timestep_emb_type="positional"))
CASES.update({
    # the config-4 network at its real size (d_model 320, 10 heads of 32, 64 tokens), conditional only, SDE noise
    "dit_full_ddpm_w1": dict(
        net=("DiT1d", dict(in_dim=29, emb_dim=128, d_model=320, n_heads=10, depth=2, timestep_emb_type="fourier")),
        x_shape=(64, 29), batch=2, fix_first_token=True, clip=3.0,
        cond=("MLPCondition", dict(in_dim=1, out_dim=128, hidden_dims=[128], act="SiLU", dropout=0.25), (1,)),
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=20, predict_noise=True)),
        sample=dict(solver="ddpm", sample_steps=3, w_cfg=1.0, temperature=0.5)),
    "dit_uncond_sde_dpmpp2m": dict(
        net=DIT_SMALL, x_shape=(12, 7), batch=5, clip=2.0,
        solver=("ContinuousDiffusionSDE", dict(predict_noise=False)),
        sample=dict(solver="sde_dpmsolver++_2M", sample_steps=6)),
    "dit_ddim_cfg": dict(
        net=DIT_SMALL, x_shape=(5, 7), batch=4, clip=2.0, cond=("IdentityCondition", dict(dropout=0.0), (32,)),
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=30, predict_noise=True)),
        sample=dict(solver="ddim", sample_steps=5, w_cfg=1.7)),
    "idql_obs_ddim_cfg": dict(
        net=("IDQLMlp", dict(obs_dim=11, act_dim=3, emb_dim=16, hidden_dim=64, n_blocks=2)), x_shape=(3,), batch=7,
        clip=1.0, cond=("IdentityCondition", dict(dropout=0.0), (11,)),
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=20, predict_noise=True)),
        sample=dict(solver="ddim", sample_steps=5, w_cfg=1.5)),
    # config 5 as the reference pipeline really samples it (synther_d4rl_mujoco.py): DiscreteDiffusionSDE, DDPM steps
    "newidql_ddpm": dict(
        net=("NewIDQLMlp", dict(obs_dim=0, act_dim=15, emb_dim=32, hidden_dim=100, n_blocks=3)), x_shape=(15,), batch=9,
        clip=2.0, solver=("DiscreteDiffusionSDE", dict(diffusion_steps=16, predict_noise=False)),
        sample=dict(solver="ddpm", sample_steps=16, temperature=0.8)),
    # the config-5 network at its real size (hidden 1024, 6 blocks, emb 128)
    "idql_full_edm_euler": dict(
        net=("IDQLMlp", dict(obs_dim=0, act_dim=15, emb_dim=128, hidden_dim=1024, n_blocks=6)), x_shape=(15,), batch=3,
        solver=("ContinuousEDM", dict()), sample=dict(solver="euler", sample_steps=4)),
})
BIGBATCH_NETS = ("DiT1d", "IDQLMlp", "NewIDQLMlp", "ChiTransformer")
CASES.update({
    # Diffusion Policy's transformer at its pusht size (d_model 256, 4 heads of 64, 8 decoder layers), CFG with w != 1
    "chitransformer_full_cfg": dict(
        net=("ChiTransformer", dict(act_dim=2, obs_dim=20, Ta=16, To=2, d_model=256, nhead=4, num_layers=8)),
        x_shape=(16, 2), batch=2, clip=1.0, cond=("IdentityCondition", dict(dropout=0.0), (2, 20)),
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=20, predict_noise=True)),
        sample=dict(solver="ddpm", sample_steps=3, w_cfg=1.4)),
    "chitransformer_uncond_2m": dict(
        net=("ChiTransformer", dict(act_dim=3, obs_dim=5, Ta=12, To=3, d_model=64, nhead=4, num_layers=2,
                                    timestep_emb_type="fourier")),
        x_shape=(12, 3), batch=5, clip=2.0,
        solver=("ContinuousDiffusionSDE", dict(predict_noise=False)),
        sample=dict(solver="ode_dpmsolver++_2M", sample_steps=6)),
})

# ---- legacy solver classes the dp_* / dbc_* pipelines import (DPMSolver, EDM): every sampler family, 2nd order, sample_x ----
_J8 = dict(net=("JannerUNet1d", dict(in_dim=6, model_dim=8, emb_dim=8, dim_mult=[1, 2], kernel_size=5)), x_shape=(8, 6),
           batch=2, fix_obs=4, legacy=True)
CASES.update({
    "janner_legacy_dpm_ode2": dict(_J8, clip=3.0, solver=("DPMSolver", dict(predict_noise=True, noise_schedule="cosine")),
                                   sample=dict(sampler="ode_dpm_2", sample_steps=5, temperature=0.8)),
    "janner_legacy_dpm_sdepp2": dict(_J8, clip=2.0, solver=("DPMSolver", dict(predict_noise=False)),
                                     sample=dict(sampler="sde_dpmpp_2", sample_steps=5, kappa=2.0)),
    "janner_legacy_dpm_sde1_x": dict(_J8, clip=3.0, solver=("DPMSolver", dict(predict_noise=False)), method="sample_x",
                                     sample=dict(sampler="sde_dpm_1", sample_steps=4, extra_sample_steps=3, temperature=0.7)),
    "janner_legacy_dpm_odepp2_eps": dict(_J8, clip=3.0, solver=("DPMSolver", dict(predict_noise=True)),
                                         sample=dict(sampler="ode_dpmpp_2", sample_steps=6)),
    "janner_legacy_dpm_ddim_cfg": dict(_J8, clip=3.0, cond_dim=8, solver=("DPMSolver", dict(predict_noise=True)),
                                       sample=dict(sampler="ddim", sample_steps=5, w_cfg=1.6)),
    "chiunet_legacy_dpmpp1": dict(
        net=("ChiUNet1d", dict(act_dim=2, obs_dim=20, To=2, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2])),
        x_shape=(16, 2), batch=3, clip=1.0, cond=("IdentityCondition", dict(dropout=0.0), (2, 20)), legacy=True,
        solver=("DPMSolver", dict(predict_noise=False)), sample=dict(sampler="ode_dpmpp_1", sample_steps=5, w_cfg=1.0)),
    "janner_legacy_edm_euler": dict(_J8, solver=("EDM", dict()), sample=dict(solver="euler", sample_steps=6)),
    "janner_legacy_edm_heun": dict(_J8, solver=("EDM", dict(sigma_max=20.0)), sample=dict(solver="heun", sample_steps=6)),
    "janner_legacy_edm_x": dict(_J8, solver=("EDM", dict()), method="sample_x",
                                sample=dict(solver="euler", sample_steps=4, extra_sample_steps=2)),
    "idql_legacy_edm_heun_cfg": dict(
        net=("IDQLMlp", dict(obs_dim=11, act_dim=3, emb_dim=16, hidden_dim=64, n_blocks=2)), x_shape=(3,), batch=7,
        cond=("IdentityCondition", dict(dropout=0.0), (11,)), legacy=True,
        solver=("EDM", dict()), sample=dict(solver="heun", sample_steps=5, w_cfg=1.3)),
    "dql_legacy_edm_euler": dict(
        net=("DQLMlp", dict(obs_dim=17, act_dim=6)), x_shape=(6,), batch=4, legacy=True,
        cond=("IdentityCondition", dict(dropout=0.0), (17,)),
        solver=("EDM", dict()), sample=dict(solver="euler", sample_steps=5, w_cfg=1.0)),
})

# EDM-family solvers over the GEMM-shaped transformers (dp_* pipelines offer `solver=edm` with every backbone): the executors
# evaluate the network on c_in * x (scaled copy in front of the token projection) and keep x_old for the Heun corrector
CHITF_SMALL = ("ChiTransformer", dict(act_dim=2, obs_dim=20, Ta=16, To=2, d_model=64, nhead=4, num_layers=2))
CASES.update({
    "dit_edm_heun_cfg": dict(
        net=DIT_SMALL, x_shape=(5, 7), batch=4, cond=("IdentityCondition", dict(dropout=0.0), (32,)),
        solver=("ContinuousEDM", dict()), sample=dict(solver="heun", sample_steps=5, w_cfg=1.4)),
    "chitf_edm_euler_x": dict(
        net=CHITF_SMALL, x_shape=(16, 2), batch=3, cond=("IdentityCondition", dict(dropout=0.0), (2, 20)),
        solver=("ContinuousEDM", dict()), sample=dict(solver="euler", sample_steps=5, w_cfg=1.0, diffusion_x_sampling_steps=1)),
    "dit_legacy_edm_heun": dict(
        net=DIT_SMALL, x_shape=(6, 7), batch=3, cond=("IdentityCondition", dict(dropout=0.0), (32,)), legacy=True,
        solver=("EDM", dict()), sample=dict(solver="heun", sample_steps=5, w_cfg=1.0)),
    "chitf_legacy_edm_euler": dict(
        net=CHITF_SMALL, x_shape=(16, 2), batch=3, cond=("IdentityCondition", dict(dropout=0.0), (2, 20)), legacy=True,
        solver=("EDM", dict()), sample=dict(solver="euler", sample_steps=6, w_cfg=1.0)),
    "chiunet_legacy_edm_heun": dict(
        net=("ChiUNet1d", dict(act_dim=2, obs_dim=20, To=2, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2])),
        x_shape=(16, 2), batch=3, cond=("IdentityCondition", dict(dropout=0.0), (2, 20)), legacy=True,
        solver=("EDM", dict()), sample=dict(solver="heun", sample_steps=5, w_cfg=1.0)),
    "chiunet_edm_euler_cfg": dict(
        net=("ChiUNet1d", dict(act_dim=2, obs_dim=20, To=2, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2])),
        x_shape=(16, 2), batch=3, cond=("IdentityCondition", dict(dropout=0.0), (2, 20)),
        solver=("ContinuousEDM", dict()), sample=dict(solver="euler", sample_steps=5, w_cfg=1.6)),
    "dit_cm": dict(
        net=DIT_SMALL, x_shape=(5, 7), batch=4, clip=2.0, legacy=True,
        solver=("ContinuousConsistencyModel", dict()), sample=dict(sample_steps=3, temperature=0.9)),
})

# ChiUNet1d with classifier-free guidance on a doubled batch (zero observations for the unconditional half), bias-only FiLM
CASES.update({
    "chiunet_cfg_w18_ddim": dict(
        net=("ChiUNet1d", dict(act_dim=3, obs_dim=5, To=2, model_dim=32, emb_dim=32, dim_mult=[1, 2], kernel_size=3)),
        x_shape=(8, 3), batch=5, clip=1.5, cond=("IdentityCondition", dict(dropout=0.0), (2, 5)),
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=30, predict_noise=True)),
        sample=dict(solver="ddim", sample_steps=5, w_cfg=1.8)),
    "chiunet_nofilmscale_sde": dict(
        net=("ChiUNet1d", dict(act_dim=2, obs_dim=4, To=1, model_dim=32, emb_dim=16, dim_mult=[1, 2, 2], kernel_size=5,
                               cond_predict_scale=False)),
        x_shape=(16, 2), batch=3, clip=2.0, cond=("IdentityCondition", dict(dropout=0.0), (1, 4)),
        solver=("ContinuousDiffusionSDE", dict(predict_noise=False)),
        sample=dict(solver="sde_dpmsolver++_1", sample_steps=4, w_cfg=1.0)),
})

# remaining reference backbones: DVInvMlp runs on the batch-tiled MLP program, the other two on PyTorch
CASES.update({
    "dvinv_ddpm": dict(
        net=("DVInvMlp", dict(obs_dim=7, act_dim=3, emb_dim=16, hidden_dim=128)), x_shape=(3,), batch=6, clip=1.0,
        cond=("IdentityCondition", dict(dropout=0.0), (14,)),
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=10, predict_noise=True)),
        sample=dict(solver="ddpm", sample_steps=5, w_cfg=1.0)),
    "sfbc_ode1": dict(
        net=("SfBCUNet", dict(act_dim=6, emb_dim=32, hidden_dims=[64, 32, 16])), x_shape=(6,), batch=4, clip=2.0,
        cond=("IdentityCondition", dict(dropout=0.0), (32,)),
        solver=("ContinuousDiffusionSDE", dict(predict_noise=True)),
        sample=dict(solver="ode_dpmsolver_1", sample_steps=5, w_cfg=1.3)),
    "pearcetf_ddim": dict(
        net=("PearceTransformer", dict(act_dim=4, To=2, emb_dim=32, trans_emb_dim=16, nhead=4)), x_shape=(4,), batch=5,
        clip=1.0, cond=("IdentityCondition", dict(dropout=0.0), (2, 32)),
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=20, predict_noise=False)),
        sample=dict(solver="ddim", sample_steps=4, w_cfg=1.0)),
})

# ---- classifier guidance at every step (w_cg > 0: what every shipped Diffuser configuration runs) ----
CASES.update({
    "janner_cfg2_guided_ddpm": dict(
        net=JANNER_CFG2, horizon=32, batch=4, fix_obs=17, classifier=dict(kernel_size=3),
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=20, predict_noise=False)),
        sample=dict(solver="ddpm", sample_steps=5, temperature=0.5, w_cg=0.3)),
    "janner_h4_guided_eps": dict(
        net=JANNER_H4, horizon=4, batch=5, fix_obs=17, classifier=dict(kernel_size=3), clip=3.0,
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=20, predict_noise=True)),
        sample=dict(solver="ddim", sample_steps=6, w_cg=0.05)),
})

# ---- rectified flow and consistency models (Euler transport as linear records; step kind 7) ----
CASES.update({
    "janner_rflow_discrete": dict(_J8, clip=2.0, solver=("DiscreteRectifiedFlow", dict(diffusion_steps=20)),
                                  sample=dict(sample_steps=5, temperature=0.8, diffusion_x_sampling_steps=1)),
    "janner_rflow_cont_cfg": dict(_J8, cond_dim=8, solver=("ContinuousRectifiedFlow", dict()),
                                  sample=dict(sample_steps=6, w_cfg=1.5)),
    "idql_rflow_cont": dict(
        net=("IDQLMlp", dict(obs_dim=0, act_dim=15, emb_dim=32, hidden_dim=64, n_blocks=2)), x_shape=(15,), batch=6,
        clip=2.0, legacy=True, solver=("ContinuousRectifiedFlow", dict()), sample=dict(sample_steps=5, temperature=0.9)),
    "janner_cm": dict(_J8, clip=2.0, solver=("ContinuousConsistencyModel", dict()),
                      sample=dict(sample_steps=4, diffusion_x_sampling_steps=1, temperature=0.9)),
    "idql_cm_cond": dict(
        net=("IDQLMlp", dict(obs_dim=11, act_dim=3, emb_dim=16, hidden_dim=64, n_blocks=2)), x_shape=(3,), batch=7,
        cond=("IdentityCondition", dict(dropout=0.0), (11,)), legacy=True,
        solver=("ContinuousConsistencyModel", dict(sigma_max=10.0)), sample=dict(sample_steps=3)),
})

# one small case per solver (discrete + continuous) so every update rule is pinned
for _s in _ALL_SOLVERS:
    CASES[f"janner_tiny_disc_{_s}"] = dict(
        net=JANNER_TINY, horizon=8, batch=2, fix_obs=4, clip=(3.0 if _s != "ddim" else None),
        solver=("DiscreteDiffusionSDE", dict(diffusion_steps=50, predict_noise=(_s != "ddim"))),
        sample=dict(solver=_s, sample_steps=5, temperature=0.8))
    CASES[f"janner_tiny_cont_{_s}"] = dict(
        net=JANNER_TINY, horizon=8, batch=2, clip=(2.5 if _s == "ddim" else None),
        solver=("ContinuousDiffusionSDE", dict(predict_noise=(_s == "ddim"))),
        sample=dict(solver=_s, sample_steps=4))


def lib_namespace(kind: str):
    """'amd' -> this repo's mirrors; 'reference' -> the real reference (build container only)."""
    if kind == "amd":
        from cleandiffuser_amd import classifier, diffusion, nn_classifier, nn_condition, nn_diffusion
    else:
        from .ref_import import import_reference
        import_reference()
        from cleandiffuser import classifier, diffusion, nn_classifier, nn_condition, nn_diffusion
    ns = SimpleNamespace()
    for mod in (diffusion, nn_condition, nn_diffusion, classifier, nn_classifier):
        for k in dir(mod):
            if not k.startswith("_"):
                setattr(ns, k, getattr(mod, k))
    import importlib
    root = "cleandiffuser_amd" if kind == "amd" else "cleandiffuser"
    ns.DDPM = importlib.import_module(root + ".diffusion.ddpm").DDPM      # legacy class, not exported by the package
    ns.DPMSolver = importlib.import_module(root + ".diffusion.dpmsolver").DPMSolver
    ns.EDM = importlib.import_module(root + ".diffusion.edm").EDM
    mlp_mod = "cleandiffuser_amd.nn_diffusion.mlp_backbones" if kind == "amd" else "cleandiffuser.nn_diffusion.idqlmlp"
    ns.NewIDQLMlp = importlib.import_module(mlp_mod).NewIDQLMlp           # likewise (idqlmlp.py:68)
    ns.DiT1Ref = importlib.import_module(root + ".nn_diffusion.dit").DiT1Ref   # (dit.py:135; the reference package does not export it)
    return ns


def x_shape_of(c):
    return tuple(c["x_shape"]) if "x_shape" in c else (c["horizon"], c["net"][1]["in_dim"])


def make_inputs(name: str):
    c = CASES[name]
    b, xs = c["batch"], x_shape_of(c)
    prior = np.zeros((b, *xs), np.float32)
    fix_mask = None
    if c.get("fix_obs"):
        fix_mask = np.zeros(xs, np.float32)
        fix_mask[0, :c["fix_obs"]] = 1.0
        prior[:, 0, :c["fix_obs"]] = synth_array(name + "/obs", (b, c["fix_obs"]))
    if c.get("fix_first_token"):
        fix_mask = np.zeros(xs, np.float32)
        fix_mask[0] = 1.0
        prior[:, 0] = synth_array(name + "/obs", (b, xs[1]))
    n_draws = 2 * c["sample"]["sample_steps"] + c["sample"].get("diffusion_x_sampling_steps", 0) + 2
    noise = np.stack([synth_array(f"{name}/z{k}", (b, *xs)) for k in range(n_draws)])
    if c.get("cond_dim"):
        cond = synth_array(name + "/cond", (b, c["cond_dim"]))
    elif c.get("cond"):
        cond = synth_array(name + "/cond", (b, *c["cond"][2]))
    else:
        cond = None
    return dict(prior=prior, fix_mask=fix_mask, noise=noise, cond=cond)


def build(lib, name: str, device="cpu", weight_seed: int = 0):
    """-> (solver object, backbone) with deterministic synthetic weights (identical for every library)."""
    c = CASES[name]
    net_cls, net_kw = c["net"]
    net = getattr(lib, net_cls)(**net_kw)
    net.load_state_dict(synth_state_dict(net.state_dict(), weight_seed))
    cond_net = lib.IdentityCondition(dropout=0.0) if c.get("cond_dim") else None
    if c.get("cond"):
        ckw = dict(c["cond"][1])
        if isinstance(ckw.get("act"), str):
            ckw["act"] = getattr(torch.nn, ckw["act"])()
        cond_net = getattr(lib, c["cond"][0])(**ckw)
        cond_net.load_state_dict(synth_state_dict(cond_net.state_dict(), weight_seed + 2))
    inp = make_inputs(name)
    kw = dict(c["solver"][1])
    if inp["fix_mask"] is not None:
        kw["fix_mask"] = torch.from_numpy(inp["fix_mask"])
    if c.get("clip"):
        xs = x_shape_of(c)
        kw["x_max"] = torch.full((1, *xs), float(c["clip"]))
        kw["x_min"] = torch.full((1, *xs), -float(c["clip"]))
    if c.get("classifier"):
        nk = net_kw
        clf_net = lib.HalfJannerUNet1d(c["horizon"], nk["in_dim"], out_dim=1, model_dim=nk["model_dim"],
                                       emb_dim=nk["emb_dim"], dim_mult=tuple(nk["dim_mult"]), **c["classifier"])
        clf_net.load_state_dict(synth_state_dict(clf_net.state_dict(), weight_seed + 1))
        kw["classifier"] = lib.CumRewClassifier(clf_net, device=device)
    if c.get("legacy") and c["solver"][0] == "EDM":
        pass                                                   # legacy EDM has no clipping bounds
    elif c.get("legacy"):
        x_max, x_min = kw.pop("x_max", None), kw.pop("x_min", None)
        kw.update(x_max=None if x_max is None else x_max.to(device), x_min=None if x_min is None else x_min.to(device))
    agent = getattr(lib, c["solver"][0])(net, cond_net, device=device, **kw)
    agent.eval()
    return agent, net


def forward_probe(name: str, agent, inputs, device="cpu"):
    """Inputs of a stand-alone ``backbone.forward`` check with a different timestep per sample:
    -> (x, t, condition embedding | None).  Used by gen_golden (reference) and the GPU tests (this repo)."""
    c = CASES[name]
    b = c["batch"]
    x = torch.from_numpy(inputs["noise"][-1]).to(device)
    kind = c["solver"][0]
    if kind in ("DiscreteDiffusionSDE", "DDPM"):
        t = ((torch.arange(b) * 7 + 1) % c["solver"][1]["diffusion_steps"]).long().to(device)
    elif kind == "ContinuousEDM":
        t = torch.linspace(-1.0, 1.0, b).to(device)
    else:
        t = torch.linspace(0.1, 0.9, b).to(device)
    cond = None
    if inputs["cond"] is not None:
        with torch.no_grad():
            cond = agent.model_ema["condition"](torch.from_numpy(inputs["cond"]).to(device), None)
    return x, t, cond


@contextlib.contextmanager
def replay_randn(noise):
    """Make ``torch.randn_like`` return the recorded draws in order (how the reference is fed fixed noise)."""
    it = iter(noise)
    orig = torch.randn_like

    def fake(ref, *a, **k):
        z = torch.as_tensor(next(it)).to(device=ref.device, dtype=ref.dtype)
        assert z.shape == ref.shape, (z.shape, ref.shape)
        return z
    torch.randn_like = fake
    try:
        yield
    finally:
        torch.randn_like = orig


def sampler_of(agent, name: str):
    """The bound method a case calls: ``sample`` unless the case names another one (``sample_x``)."""
    return getattr(agent, CASES[name].get("method", "sample"))


def sample_kwargs(name: str, inputs, device="cpu"):
    c = CASES[name]
    kw = dict(c["sample"])
    kw["n_samples"] = c["batch"]
    if inputs["cond"] is not None:
        kw["condition_cfg"] = torch.from_numpy(inputs["cond"]).to(device)
    return kw
