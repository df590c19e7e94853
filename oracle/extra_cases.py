"""Reference fixtures for the shapes the small ``cases.CASES`` fixtures do not reach -- TEST INFRASTRUCTURE (oracle/__init__.py).

Round 1 compared these shapes (shipped widths / horizons: PearceMlp 64/192/512, long-horizon and wide JannerUNet1d, the
model_dim-64 Diffuser nets of kitchen / antmaze with classifier guidance, ChiTransformer Ta = 10, DiT1d with 10 and 40 tokens /
depth 8, ChiUNet1d at the config-3 width) against this repo's own PyTorch executor on the GPU box's CPU and had to widen the
tolerance for host-BLAS summation order.  Here every scenario is a pair of pure functions of a *library namespace*
(``cases.lib_namespace("reference" | "amd")``) and a device, so ``python -m oracle.gen_golden_extra`` (build container) runs the REAL
reference on them and commits ``tests/golden/extra_<name>.npz``; the GPU tests build the same agents from this repo's classes on
the device and are held to the 1e-4 bar like every other fixture.  Weights: ``load_synth`` (PCG64 streams keyed by parameter
name); inputs: seeded ``torch.Generator`` draws -- nothing but the outputs is stored.
"""
from typing import Callable, Dict

import torch

from cleandiffuser_amd.utils import load_synth
from . import cases


def _sample(agent, lib_kind: str, prior, zs, **kw):
    """agent.sample on recorded draws: this repo takes ``noise=[...]``, the reference gets them through ``torch.randn_like``."""
    dev = prior.device
    if lib_kind == "amd":
        return agent.sample(prior, noise=[z.to(dev) for z in zs], **kw)
    with cases.replay_randn([z for z in zs]):
        return agent.sample(prior, **kw)


# --------------------------------------------------------------------------------------------------------------------- #
def pearce(hidden: int, batch: int):
    steps = 8

    def run(lib, kind, device):
        net = load_synth(lib.PearceMlp(6, To=1, emb_dim=32, hidden_dim=hidden), 3)
        cond = load_synth(lib.PearceObsCondition(11, 32, flatten=True, dropout=0.0), 4)
        agent = lib.DiscreteDiffusionSDE(net, cond, predict_noise=False, x_max=torch.ones(1, 6), x_min=-torch.ones(1, 6),
                                         diffusion_steps=steps, device=device)
        agent.eval()
        g = torch.Generator().manual_seed(hidden)
        obs = torch.randn(batch, 1, 11, generator=g)
        zs = [torch.randn(batch, 6, generator=g) for _ in range(steps + 1)]
        x, _ = _sample(agent, kind, torch.zeros(batch, 6, device=device), zs, solver="ddpm", n_samples=batch, sample_steps=steps,
                       temperature=0.7, w_cfg=1.0, condition_cfg=obs.to(device))
        return {"x": x}
    return run


def mlp_cfg_pair(which: str):
    """Tile-MLP denoisers with a classifier-free-guidance PAIR (w_cfg not in {0, 1}: conditional and zero-condition forward per
    step, reference diffusionsde.py:175-206)."""
    steps, batch = 6, 21

    def run(lib, kind, device):
        g = torch.Generator().manual_seed(77)
        if which == "pearce":
            net = load_synth(lib.PearceMlp(6, To=1, emb_dim=32, hidden_dim=128), 5)
            cond = load_synth(lib.PearceObsCondition(11, 32, flatten=True, dropout=0.0), 6)
            obs, d = torch.randn(batch, 1, 11, generator=g), 6
        else:
            net = load_synth(lib.DQLMlp(11, 6, emb_dim=16), 5)
            cond = lib.IdentityCondition(dropout=0.0)
            obs, d = torch.randn(batch, 11, generator=g), 6
        agent = lib.DiscreteDiffusionSDE(net, cond, predict_noise=False, x_max=torch.ones(1, d), x_min=-torch.ones(1, d),
                                         diffusion_steps=steps, device=device)
        agent.eval()
        zs = [torch.randn(batch, d, generator=g) for _ in range(steps + 1)]
        x, _ = _sample(agent, kind, torch.zeros(batch, d, device=device), zs, solver="ddpm", n_samples=batch, sample_steps=steps,
                       temperature=0.8, w_cfg=1.5, condition_cfg=obs.to(device))
        return {"x": x}
    return run


def idql_wide():
    """IDQLMlp with hidden 2048 (> the 1024 the residual-MLP executor's LayerNorm used to hold): 2 blocks, EDM Euler, B = 9."""
    def run(lib, kind, device):
        net = load_synth(lib.IDQLMlp(11, 6, emb_dim=32, hidden_dim=2048, n_blocks=2), 8)
        agent = lib.ContinuousEDM(net, lib.IdentityCondition(dropout=0.0), x_max=torch.ones(1, 6), x_min=-torch.ones(1, 6), device=device)
        agent.eval()
        g = torch.Generator().manual_seed(8)
        obs = torch.randn(9, 11, generator=g)
        zs = [torch.randn(9, 6, generator=g) for _ in range(6)]
        x, _ = _sample(agent, kind, torch.zeros(9, 6, device=device), zs, solver="euler", n_samples=9, sample_steps=5,
                       condition_cfg=obs.to(device), w_cfg=1.0)
        return {"x": x}
    return run


def mlpnn():
    """MlpNNDiffusion (reference nn_diffusion/mlps.py): ReLU MLP over [x | map_noise(t) + condition], DDIM, CFG pair w = 1.4."""
    def run(lib, kind, device):
        net = load_synth(lib.MlpNNDiffusion(5, emb_dim=16, hidden_dims=[64, 128]), 12)
        agent = lib.DiscreteDiffusionSDE(net, lib.IdentityCondition(dropout=0.0), predict_noise=True, x_max=2 * torch.ones(1, 5),
                                         x_min=-2 * torch.ones(1, 5), diffusion_steps=20, device=device)
        agent.eval()
        g = torch.Generator().manual_seed(12)
        cond = torch.randn(19, 16, generator=g)
        zs = [torch.randn(19, 5, generator=g) for _ in range(6)]
        x, _ = _sample(agent, kind, torch.zeros(19, 5, device=device), zs, solver="ddim", n_samples=19, sample_steps=5, w_cfg=1.4,
                       condition_cfg=cond.to(device))
        return {"x": x}
    return run


def janner_long(horizon: int, dim_mult, model_dim: int):
    D, B, steps = 6, 3, 3

    def run(lib, kind, device):
        net = load_synth(lib.JannerUNet1d(D, model_dim=model_dim, emb_dim=32, dim_mult=dim_mult, kernel_size=5), 11)
        fm = torch.zeros(horizon, D)
        fm[0, :4] = 1.0
        agent = lib.DiscreteDiffusionSDE(net, None, fix_mask=fm, diffusion_steps=10, predict_noise=False, device=device)
        agent.eval()
        g = torch.Generator().manual_seed(horizon)
        prior = torch.zeros(B, horizon, D)
        prior[:, 0, :4] = torch.randn(B, 4, generator=g)
        zs = [torch.randn(B, horizon, D, generator=g) for _ in range(steps + 1)]
        x, _ = _sample(agent, kind, prior.to(device), zs, solver="ddim", n_samples=B, sample_steps=steps, temperature=0.8)
        return {"x": x, "_agent": agent}
    return run


def shipped_diffuser(size: str):
    """kitchen (H 32, D 69) / antmaze (H 64, D 37), model_dim 64, CumRewClassifier(HalfJannerUNet1d): stand-alone forward,
    unguided loop, guided loop (w_cg 0.2) + the classifier's log_p."""
    H, D, n_obs = (32, 69, 60) if size == "kitchen" else (64, 37, 29)
    B, steps = 3, 3

    def run(lib, kind, device):
        net = load_synth(lib.JannerUNet1d(D, model_dim=64, emb_dim=64, dim_mult=[1, 2, 2, 2], kernel_size=5), 21)
        clf_net = load_synth(lib.HalfJannerUNet1d(H, D, out_dim=1, model_dim=64, emb_dim=64, dim_mult=(1, 2, 2, 2), kernel_size=3), 22)
        fm = torch.zeros(H, D)
        fm[0, :n_obs] = 1.0
        agent = lib.DiscreteDiffusionSDE(net, None, fix_mask=fm, classifier=lib.CumRewClassifier(clf_net, device=device),
                                         diffusion_steps=10, predict_noise=False, device=device)
        agent.eval()
        agent.classifier.eval()
        g = torch.Generator().manual_seed(5)
        prior = torch.zeros(B, H, D)
        prior[:, 0, :n_obs] = torch.randn(B, n_obs, generator=g)
        zs = [torch.randn(B, H, D, generator=g) for _ in range(steps + 2)]
        t = torch.tensor([1, 4, 8])
        with torch.no_grad():
            fwd = agent.model_ema["diffusion"](zs[0].to(device), t.to(device), None)
        kw = dict(solver="ddpm", n_samples=B, sample_steps=steps, temperature=0.5)
        x, _ = _sample(agent, kind, prior.to(device), zs, w_cg=0.0, **kw)
        xg, log = _sample(agent, kind, prior.to(device), zs, w_cg=0.2, condition_cg=None, **kw)
        return {"fwd": fwd, "x": x, "x_guided": xg, "log_p": log["log_p"], "_agent": agent}
    return run


def transformer(which: str):
    B, steps = 3, 3

    def run(lib, kind, device):
        if which == "chitf_ta10":
            net, x_shape, cond_shape = lib.ChiTransformer(7, 23, 10, 2, d_model=256, nhead=4, num_layers=3), (10, 7), (2, 23)
        elif which == "chitf_enc2":           # transformer condition encoder (n_cond_layers > 0, reference chitransformer.py:91-95)
            net = lib.ChiTransformer(7, 23, 10, 2, d_model=128, nhead=4, num_layers=2, n_cond_layers=2)
            x_shape, cond_shape = (10, 7), (2, 23)
        elif which == "dit_h96":              # more than 64 tokens: streamed-key attention kernel
            net, x_shape, cond_shape = lib.DiT1d(7, emb_dim=64, d_model=128, n_heads=4, depth=2), (96, 7), (64,)
        elif which == "dit_h10_d384":
            net, x_shape, cond_shape = lib.DiT1d(7, emb_dim=64, d_model=384, n_heads=6, depth=2), (10, 7), (64,)
        else:
            net = lib.DiT1d(29, emb_dim=128, d_model=256, n_heads=8, depth=8, timestep_emb_type="fourier")
            x_shape, cond_shape = (40, 29), (128,)
        lim = 50.0 if which == "chitf_enc2" else 2.0          # (synthetic weights saturate a +-2 clip on most elements)
        agent = lib.DiscreteDiffusionSDE(load_synth(net, 31), lib.IdentityCondition(dropout=0.0), predict_noise=True,
                                         x_max=lim * torch.ones(1, *x_shape), x_min=-lim * torch.ones(1, *x_shape), diffusion_steps=20,
                                         device=device)
        agent.eval()
        g = torch.Generator().manual_seed(len(which))
        cond = torch.randn(B, *cond_shape, generator=g)
        zs = [torch.randn(B, *x_shape, generator=g) for _ in range(steps + 1)]
        x, _ = _sample(agent, kind, torch.zeros(B, *x_shape, device=device), zs, solver="ddim", n_samples=B, sample_steps=steps,
                       w_cfg=1.3, condition_cfg=cond.to(device))
        return {"x": x}
    return run


def chiunet_cfg3_width():
    """ChiUNet1d at the BASELINE config-3 width (model_dim 256, 68.9 M parameters), legacy DDPM, 4 steps, B = 2."""
    B, steps = 2, 4

    def run(lib, kind, device):
        net = load_synth(lib.ChiUNet1d(2, 20, 2, model_dim=256, emb_dim=256, dim_mult=[1, 2, 2], obs_as_global_cond=True), 41)
        agent = lib.DDPM(net, lib.IdentityCondition(dropout=0.0), diffusion_steps=steps, x_max=torch.ones(1, 16, 2, device=device),
                         x_min=-torch.ones(1, 16, 2, device=device), device=device)
        agent.eval()
        g = torch.Generator().manual_seed(41)
        cond = torch.randn(B, 2, 20, generator=g)
        zs = [torch.randn(B, 16, 2, generator=g) for _ in range(steps + 1)]
        x, _ = _sample(agent, kind, torch.zeros(B, 16, 2, device=device), zs, n_samples=B, sample_steps=steps,
                       condition_cfg=cond.to(device), w_cfg=1.0)
        return {"x": x, "_agent": agent}
    return run


def chiunet_local_cond():
    """ChiUNet1d with LOCAL conditioning (obs_as_global_cond=False, reference nn_diffusion/chiunet.py:78-82, 153-185): one observation
    row per action position, two extra residual blocks whose outputs join the first down level and the last up level."""
    B, steps, Ta = 3, 4, 16

    def run(lib, kind, device):
        net = load_synth(lib.ChiUNet1d(2, 5, Ta, model_dim=64, emb_dim=64, dim_mult=[1, 2, 2], obs_as_global_cond=False), 43)
        agent = lib.DDPM(net, lib.IdentityCondition(dropout=0.0), diffusion_steps=steps, x_max=torch.ones(1, Ta, 2, device=device),
                         x_min=-torch.ones(1, Ta, 2, device=device), device=device)
        agent.eval()
        g = torch.Generator().manual_seed(43)
        cond = torch.randn(B, Ta, 5, generator=g)
        zs = [torch.randn(B, Ta, 2, generator=g) for _ in range(steps + 1)]
        with torch.no_grad():
            fwd = agent.model_ema["diffusion"](zs[0].to(device), torch.tensor([0, 1, 3], device=device), cond.to(device))
        x, _ = _sample(agent, kind, torch.zeros(B, Ta, 2, device=device), zs, n_samples=B, sample_steps=steps,
                       condition_cfg=cond.to(device), w_cfg=1.0)
        return {"fwd": fwd, "x": x, "_agent": agent}
    return run


SCENARIOS: Dict[str, Callable] = {
    "pearce_h64": pearce(64, 5), "pearce_h192": pearce(192, 37), "pearce_h512": pearce(512, 16),
    "janner_h128": janner_long(128, [1, 2, 2, 2], 32), "janner_h64_w48": janner_long(64, [1, 4, 2], 48),
    "diffuser_kitchen": shipped_diffuser("kitchen"), "diffuser_antmaze": shipped_diffuser("antmaze"),
    "chitf_ta10": transformer("chitf_ta10"), "chitf_enc2": transformer("chitf_enc2"), "dit_h96": transformer("dit_h96"), "dit_h10_d384": transformer("dit_h10_d384"), "dit_h40_depth8": transformer("dit_h40_depth8"),
    "chiunet_cfg3_width": chiunet_cfg3_width(), "chiunet_local_cond": chiunet_local_cond(),
    "pearce_cfg_pair": mlp_cfg_pair("pearce"), "dql_cfg_pair": mlp_cfg_pair("dql"), "idql_h2048": idql_wide(),
    "mlpnn_cfg_pair": mlpnn(),
}


# --------------------------------------------------------------------------------------------------------------------- #
# BASELINE.json's five configurations at their EXACT (network size x solver x step count x guidance) combination, small batch
# (construction and call: BASELINE.md section 2).  Round 2 pinned these nets only at reduced width or reduced step count.
def baseline_config(which: str, batch: int, variant: str = ""):
    """`variant` "tied" (config 4): the synthetic DiT1d behaves like a TRAINED noise predictor -- its output layer is tied to the
    input projection (final_layer.linear.weight = 0.25 x_proj.weight^T, its bias x 0.1), so the predicted noise tracks x_t and the
    un-clipped 10-step DPM-Solver++ result stays |x| <= 6.2 (plain synthetic weights: the first step divides by alpha(1) = 0.0066
    and the result reaches |x| = 589, which leaves a relative comparison no resolving power -- VERDICT r3 weak #1).  Same network
    size, solver, step count and guidance; held to an ABSOLUTE 1e-4.  `variant` "d27" (config 5): the real hopper transition width
    2 * 11 + 3 + 2 = 27 instead of BASELINE.json's 15 (report both)."""
    def run(lib, kind, device):
        g = torch.Generator().manual_seed(1000 + len(which) + batch)
        B = batch
        if which == "cfg1":         # PearceMlp DBC (tutorials/1_...py:60-81,138-141): 100-step DDPM, w_cfg = 1
            net = load_synth(lib.PearceMlp(6, To=1, emb_dim=64, hidden_dim=256), 51)
            cond = load_synth(lib.PearceObsCondition(17, 64, flatten=True, dropout=0.0), 52)
            agent = lib.DiscreteDiffusionSDE(net, cond, predict_noise=False, x_max=torch.ones(1, 6), x_min=-torch.ones(1, 6),
                                             diffusion_steps=100, device=device)
            agent.eval()
            obs = torch.randn(B, 1, 17, generator=g)
            zs = [torch.randn(B, 6, generator=g) for _ in range(101)]
            x, _ = _sample(agent, kind, torch.zeros(B, 6, device=device), zs, solver="ddpm", n_samples=B, sample_steps=100,
                           temperature=0.5, w_cfg=1.0, condition_cfg=obs.to(device))
        elif which in ("cfg2", "cfg2_guided"):   # JannerUNet1d Diffuser, H = 32, D = 23, 20-step DDIM (north star)
            net = load_synth(lib.JannerUNet1d(23, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5), 53)
            fm = torch.zeros(32, 23)
            fm[0, :17] = 1.0
            clf = None
            if which == "cfg2_guided":           # what diffuser_d4rl_mujoco.py runs: DDPM steps, CumRewClassifier guidance, log_p
                cn = load_synth(lib.HalfJannerUNet1d(32, 23, out_dim=1, model_dim=32, emb_dim=32, dim_mult=(1, 2, 2, 2),
                                                     kernel_size=3), 54)
                clf = lib.CumRewClassifier(cn, device=device)
            agent = lib.DiscreteDiffusionSDE(net, None, fix_mask=fm, classifier=clf, diffusion_steps=20, predict_noise=False,
                                             device=device)
            agent.eval()
            if clf is not None:
                clf.eval()
            prior = torch.zeros(B, 32, 23)
            prior[:, 0, :17] = torch.randn(B, 17, generator=g)
            zs = [torch.randn(B, 32, 23, generator=g) for _ in range(21)]
            if which == "cfg2":
                x, _ = _sample(agent, kind, prior.to(device), zs[:1], solver="ddim", n_samples=B, sample_steps=20, temperature=0.5)
            else:
                x, log = _sample(agent, kind, prior.to(device), zs, solver="ddpm", n_samples=B, sample_steps=20, temperature=0.5,
                                 w_cg=0.1, condition_cg=None)
                return {"x": x, "log_p": log["log_p"]}
        elif which == "cfg3":       # ChiUNet1d Diffusion Policy (dp_pusht), model_dim 256, legacy DDPM class, 50 steps
            net = load_synth(lib.ChiUNet1d(2, 20, 2, model_dim=256, emb_dim=256, dim_mult=[1, 2, 2], obs_as_global_cond=True), 55)
            agent = lib.DDPM(net, lib.IdentityCondition(dropout=0.0), diffusion_steps=50, x_max=torch.ones(1, 16, 2, device=device),
                             x_min=-torch.ones(1, 16, 2, device=device), device=device)
            agent.eval()
            cond = torch.randn(B, 2, 20, generator=g)
            zs = [torch.randn(B, 16, 2, generator=g) for _ in range(51)]
            x, _ = _sample(agent, kind, torch.zeros(B, 16, 2, device=device), zs, n_samples=B, sample_steps=50,
                           condition_cfg=cond.to(device), w_cfg=1.0)
        elif which == "cfg4":       # DiT1d Decision Diffuser: d 320 / 10 heads / 64 tokens, CFG w = 2, 10-step DPM-Solver++ 2M
            net = load_synth(lib.DiT1d(29, emb_dim=128, d_model=320, n_heads=10, depth=2, timestep_emb_type="fourier"), 56)
            if variant == "tied":
                with torch.no_grad():
                    sd = dict(net.named_parameters())
                    sd["final_layer.linear.weight"].copy_(0.25 * sd["x_proj.weight"].t())
                    sd["final_layer.linear.bias"].mul_(0.1)
            cond = load_synth(lib.MLPCondition(1, 128, [128], torch.nn.SiLU(), dropout=0.25), 57)
            fm = torch.zeros(64, 29)
            fm[0] = 1.0
            agent = lib.ContinuousDiffusionSDE(net, cond, fix_mask=fm, predict_noise=True, noise_schedule="linear", device=device)
            agent.eval()
            prior = torch.zeros(B, 64, 29)
            prior[:, 0] = torch.randn(B, 29, generator=g)
            zs = [torch.randn(B, 64, 29, generator=g)]
            ccfg = 0.3 * torch.ones(B, 1) + 0.1 * torch.randn(B, 1, generator=g)
            if _ROWS is not None:                # (a same-box yardstick on a subset of the batch: trajectories are independent)
                prior, zs, ccfg = prior[_ROWS], [z[_ROWS] for z in zs], ccfg[_ROWS]
                B = prior.shape[0]
            x, _ = _sample(agent, kind, prior.to(device), zs, solver="ode_dpmsolver++_2M", n_samples=B, sample_steps=10,
                           condition_cfg=ccfg.to(device), w_cfg=2.0, temperature=0.5)
        else:                       # cfg5: SynthER ResidualMLP = IDQLMlp 1024 x 6, 128-step EDM Euler
            D = 27 if variant == "d27" else 15
            net = load_synth(lib.IDQLMlp(0, D, emb_dim=128, hidden_dim=1024, n_blocks=6), 58)
            agent = lib.ContinuousEDM(net, None, device=device)
            agent.eval()
            zs = [torch.randn(B, D, generator=g)]
            if _ROWS is not None:                # (the fixture of a batch the CPU cannot afford: samples are independent, newedm.py:286-438)
                zs, B = [z[_ROWS] for z in zs], len(_ROWS)
            x, _ = _sample(agent, kind, torch.zeros(B, D, device=device), zs, solver="euler", n_samples=B, sample_steps=128)
        return {"x": x}
    return run


SCENARIOS.update({
    "baseline_cfg1": baseline_config("cfg1", 5), "baseline_cfg2_b256": baseline_config("cfg2", 256),
    "baseline_cfg2_guided": baseline_config("cfg2_guided", 8), "baseline_cfg3": baseline_config("cfg3", 2),
    "baseline_cfg4": baseline_config("cfg4", 3), "baseline_cfg5": baseline_config("cfg5", 3),
    "baseline_cfg4_tied": baseline_config("cfg4", 3, "tied"), "baseline_cfg5_d27": baseline_config("cfg5", 3, "d27"),
})

# The same configurations at a batch that crosses the GEMM executors' tile / chunk boundaries (full-size behaviour
# pinned against the REFERENCE, not against this repo's CPU executor).  Fixtures only (a few TFLOP of CPU work each in the build
# container); the CPU suite does not re-run them, the GPU suite holds the native path to them at 1e-4.
GPU_ONLY: Dict[str, Callable] = {
    "baseline_cfg3_b130": baseline_config("cfg3", 130), "baseline_cfg4_tied_b96": baseline_config("cfg4", 96, "tied"),
    "baseline_cfg5_b300": baseline_config("cfg5", 300), "baseline_cfg5_d27_b300": baseline_config("cfg5", 300, "d27"),
    # round 5: config 4 at its EXACT per-GPU shard (B = 4096 over 8 GPUs -> 512 trajectories = 65 536 token rows with the CFG pair)
    "baseline_cfg4_tied_b512": baseline_config("cfg4", 512, "tied"),
    # round 6: config 5 ACROSS the executor's real chunk boundary -- cdx_resmlp_run cuts a call into 16 384-row chunks (what one rank of
    # the 8-GPU run does eight times per call); the fixture holds the reference's result for the rows either side of the cut
    "baseline_cfg5_b16684": baseline_config("cfg5", 16384 + 300),
    # round 6: the classifier-guided Diffuser loop at a batch the GROUPED guided program serves (128 < B <= 256; 200 = 50 groups of four)
    "baseline_cfg2_guided_b200": baseline_config("cfg2_guided", 200),
}
# Fixtures of these scenarios hold the listed ROWS only (the reference sampled just those: samples are independent; the device run is the whole batch)
ROW_SUBSET = {"baseline_cfg5_b16684": [0, 1, 255, 256, 8191, 16382, 16383, 16384, 16385, 16511, 16512, 16683]}
# Fixtures of these scenarios keep every STRIDE-th trajectory only (trajectories are independent; the device run is the whole batch)
SUBSAMPLED = {"baseline_cfg4_tied_b512": 4, "baseline_cfg2_guided_b200": 3}


# --------------------------------------------------------------------------------------------------------------------- #
# row f3: the two backbones that were PyTorch mirrors until round 3 (reference pearcetransformer.py:91-151, dit.py:135-180)
def pearce_transformer(default_size: bool):
    """PearceTransformer: stand-alone forward (per-sample timesteps), w_cfg = 1 loop and a CFG pair (w_cfg = 1.6).  `default_size`:
    the constructor defaults the DBC pipelines use (emb 128, trans_emb_dim 64, 16 heads -> 1024-wide attention), To = 2."""
    B, steps = 5, 4

    def run(lib, kind, device):
        kw = dict(To=2, emb_dim=128, trans_emb_dim=64, nhead=16) if default_size else dict(To=3, emb_dim=32, trans_emb_dim=16, nhead=4)
        net = load_synth(lib.PearceTransformer(6, **kw), 71)
        agent = lib.DiscreteDiffusionSDE(net, lib.IdentityCondition(dropout=0.0), predict_noise=False, x_max=torch.ones(1, 6),
                                         x_min=-torch.ones(1, 6), diffusion_steps=20, device=device)
        agent.eval()
        g = torch.Generator().manual_seed(71)
        cond = torch.randn(B, kw["To"], kw["emb_dim"], generator=g)
        zs = [torch.randn(B, 6, generator=g) for _ in range(steps + 1)]
        with torch.no_grad():
            fwd = agent.model_ema["diffusion"](zs[0].to(device), torch.tensor([0, 3, 7, 11, 19], device=device), cond.to(device))
        skw = dict(solver="ddpm", n_samples=B, sample_steps=steps, temperature=0.7, condition_cfg=cond.to(device))
        x1, _ = _sample(agent, kind, torch.zeros(B, 6, device=device), zs, w_cfg=1.0, **skw)
        x2, _ = _sample(agent, kind, torch.zeros(B, 6, device=device), zs, w_cfg=1.6, **skw)
        return {"fwd": fwd, "x": x1, "x_cfg": x2}
    return run


def dit1ref():
    """DiT1Ref: state rows [reference | noisy] with the reference half held by the fix-mask; stand-alone forward, CFG pair DDIM loop."""
    B, steps, T, D = 3, 4, 12, 5

    def run(lib, kind, device):
        net = load_synth(lib.DiT1Ref(D, emb_dim=32, d_model=64, n_heads=4, depth=2), 73)
        fm = torch.zeros(T, 2 * D)
        fm[:, :D] = 1.0
        lim = 2.0 * torch.ones(1, T, 2 * D)
        agent = lib.DiscreteDiffusionSDE(net, lib.IdentityCondition(dropout=0.0), fix_mask=fm, predict_noise=True, x_max=lim, x_min=-lim,
                                         diffusion_steps=20, device=device)
        agent.eval()
        g = torch.Generator().manual_seed(73)
        prior = torch.zeros(B, T, 2 * D)
        prior[:, :, :D] = torch.randn(B, T, D, generator=g)
        cond = torch.randn(B, 32, generator=g)
        zs = [torch.randn(B, T, 2 * D, generator=g) for _ in range(steps + 1)]
        with torch.no_grad():
            fwd = agent.model_ema["diffusion"](zs[0].to(device), torch.tensor([1, 8, 15], device=device), cond.to(device))
        x, _ = _sample(agent, kind, prior.to(device), zs, solver="ddim", n_samples=B, sample_steps=steps, w_cfg=1.5,
                       condition_cfg=cond.to(device))
        xu, _ = _sample(agent, kind, prior.to(device), zs, solver="sde_dpmsolver++_1", n_samples=B, sample_steps=steps, w_cfg=0.0)
        return {"fwd": fwd, "x": x, "x_uncond": xu}
    return run


SCENARIOS.update({"pearcetf_small": pearce_transformer(False), "pearcetf_default": pearce_transformer(True), "dit1ref": dit1ref()})


def janner_attention():
    """JannerUNet1d(attention=True) -- what the reference's own tests/test_janner_unet.py instantiates: LinearAttention after every
    level's second block and between the middle blocks (reference jannerunet.py:72-95, 124-152).  Stand-alone forward (per-sample
    timesteps) and an unconditional 4-step DDIM loop."""
    B, H, D, steps = 3, 16, 6, 4

    def run(lib, kind, device):
        net = load_synth(lib.JannerUNet1d(D, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2], kernel_size=5, attention=True), 81)
        fm = torch.zeros(H, D)
        fm[0, :4] = 1.0
        agent = lib.DiscreteDiffusionSDE(net, None, fix_mask=fm, diffusion_steps=10, predict_noise=False, device=device)
        agent.eval()
        g = torch.Generator().manual_seed(81)
        prior = torch.zeros(B, H, D)
        prior[:, 0, :4] = torch.randn(B, 4, generator=g)
        zs = [torch.randn(B, H, D, generator=g) for _ in range(steps + 1)]
        with torch.no_grad():
            fwd = agent.model_ema["diffusion"](zs[0].to(device), torch.tensor([0, 4, 9], device=device), None)
        x, _ = _sample(agent, kind, prior.to(device), zs, solver="ddim", n_samples=B, sample_steps=steps, temperature=0.8)
        return {"fwd": fwd, "x": x}
    return run


SCENARIOS["janner_attention"] = janner_attention()


def janner_attention_conditional():
    """JannerUNet1d(attention=True) WITH a condition embedding (reference jannerunet.py:160-164: emb = map_noise(t) + condition): the
    stand-alone forward, a w_cfg = 1 DDIM loop and a classifier-free-guidance pair (w_cfg = 1.5)."""
    B, H, D, steps = 3, 16, 6, 4

    def run(lib, kind, device):
        net = load_synth(lib.JannerUNet1d(D, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2], kernel_size=5, attention=True), 82)
        agent = lib.DiscreteDiffusionSDE(net, lib.IdentityCondition(dropout=0.0), diffusion_steps=10, predict_noise=True,
                                         x_max=2 * torch.ones(1, H, D), x_min=-2 * torch.ones(1, H, D), device=device)
        agent.eval()
        g = torch.Generator().manual_seed(82)
        cond = 0.5 * torch.randn(B, 32, generator=g)
        zs = [torch.randn(B, H, D, generator=g) for _ in range(steps + 1)]
        with torch.no_grad():
            fwd = agent.model_ema["diffusion"](zs[0].to(device), torch.tensor([1, 4, 9], device=device), cond.to(device))
        kw = dict(solver="ddim", n_samples=B, sample_steps=steps, temperature=0.8, condition_cfg=cond.to(device))
        x1, _ = _sample(agent, kind, torch.zeros(B, H, D, device=device), zs, w_cfg=1.0, **kw)
        x2, _ = _sample(agent, kind, torch.zeros(B, H, D, device=device), zs, w_cfg=1.5, **kw)
        return {"fwd": fwd, "x_w1": x1, "x_pair": x2}
    return run


SCENARIOS["janner_attention_conditional"] = janner_attention_conditional()


def chitransformer_pusht_full():
    """Diffusion Policy's transformer at the dp_pusht size AND step count (tools/bench_configs.py cfgT: d_model 256, 4 heads, 8 decoder
    layers, Ta = 16, To = 2, obs 20; 100-step DDPM over DiscreteDiffusionSDE, w_cfg = 1), B = 2."""
    B, steps = 2, 100

    def run(lib, kind, device):
        net = load_synth(lib.ChiTransformer(2, 20, 16, 2, d_model=256, nhead=4, num_layers=8), 91)
        one = torch.ones(1, 16, 2)
        agent = lib.DiscreteDiffusionSDE(net, lib.IdentityCondition(dropout=0.0), predict_noise=True, x_max=one, x_min=-one,
                                         diffusion_steps=steps, device=device)
        agent.eval()
        g = torch.Generator().manual_seed(91)
        obs = torch.randn(B, 2, 20, generator=g)
        zs = [torch.randn(B, 16, 2, generator=g) for _ in range(steps + 1)]
        x, _ = _sample(agent, kind, torch.zeros(B, 16, 2, device=device), zs, solver="ddpm", n_samples=B, sample_steps=steps,
                       condition_cfg=obs.to(device), w_cfg=1.0)
        return {"x": x}
    return run


def shipped_diffuser_full_steps(size: str):
    """The shipped kitchen / antmaze Diffuser sizes at the shipped sampling setting: 20-step DDPM with classifier guidance and the final
    log_p (configs/diffuser/{kitchen,antmaze}: solver ddpm, sampling_steps 20), B = 3 -- `shipped_diffuser` above stops after 3 steps."""
    H, D, n_obs = (32, 69, 60) if size == "kitchen" else (64, 37, 29)
    B, steps = 3, 20

    def run(lib, kind, device):
        net = load_synth(lib.JannerUNet1d(D, model_dim=64, emb_dim=64, dim_mult=[1, 2, 2, 2], kernel_size=5), 21)
        clf_net = load_synth(lib.HalfJannerUNet1d(H, D, out_dim=1, model_dim=64, emb_dim=64, dim_mult=(1, 2, 2, 2), kernel_size=3), 22)
        fm = torch.zeros(H, D)
        fm[0, :n_obs] = 1.0
        agent = lib.DiscreteDiffusionSDE(net, None, fix_mask=fm, classifier=lib.CumRewClassifier(clf_net, device=device),
                                         diffusion_steps=steps, predict_noise=False, device=device)
        agent.eval()
        agent.classifier.eval()
        g = torch.Generator().manual_seed(6)
        prior = torch.zeros(B, H, D)
        prior[:, 0, :n_obs] = torch.randn(B, n_obs, generator=g)
        zs = [torch.randn(B, H, D, generator=g) for _ in range(steps + 1)]
        xg, log = _sample(agent, kind, prior.to(device), zs, solver="ddpm", n_samples=B, sample_steps=steps, temperature=0.5, w_cg=0.05,
                          condition_cg=None)
        return {"x_guided": xg, "log_p": log["log_p"]}
    return run


SCENARIOS.update({"chitf_pusht_full": chitransformer_pusht_full(), "diffuser_kitchen_20": shipped_diffuser_full_steps("kitchen"),
                  "diffuser_antmaze_20": shipped_diffuser_full_steps("antmaze")})


def edm_classifier_guidance():
    """Classifier guidance under ContinuousEDM (reference newedm.py:217-284): D + w sigma^2 grad at every network evaluation of a Heun
    loop (predictor and corrector), the classifier fed (x_t, ln(sigma) / 4); with condition_cg = None the reference applies NO shift but
    still scores the result -- both variants, plus the final log_p at sigma_min."""
    B, H, D, steps = 3, 8, 6, 4

    def run(lib, kind, device):
        net = load_synth(lib.JannerUNet1d(D, model_dim=16, emb_dim=16, dim_mult=[1, 2], kernel_size=5), 95)
        clf_net = load_synth(lib.HalfJannerUNet1d(H, D, out_dim=1, model_dim=16, emb_dim=16, dim_mult=(1, 2), kernel_size=3), 96)
        fm = torch.zeros(H, D)
        fm[0, :4] = 1.0
        lim = 2.0 * torch.ones(1, H, D)
        agent = lib.ContinuousEDM(net, None, fix_mask=fm, classifier=lib.CumRewClassifier(clf_net, device=device), x_max=lim, x_min=-lim,
                                  device=device)
        agent.eval()
        agent.classifier.eval()
        g = torch.Generator().manual_seed(95)
        prior = torch.zeros(B, H, D)
        prior[:, 0, :4] = torch.randn(B, 4, generator=g)
        zs = [torch.randn(B, H, D, generator=g)]
        kw = dict(n_samples=B, sample_steps=steps)
        xg, lg = _sample(agent, kind, prior.to(device), zs, solver="heun", w_cg=0.3, condition_cg=torch.ones(B, 1, device=device), **kw)
        xe, le = _sample(agent, kind, prior.to(device), zs, solver="euler", w_cg=0.5, condition_cg=torch.ones(B, 1, device=device),
                         diffusion_x_sampling_steps=1, **kw)
        xn, ln = _sample(agent, kind, prior.to(device), zs, solver="heun", w_cg=0.3, condition_cg=None, **kw)
        return {"x_heun": xg, "logp_heun": lg["log_p"], "x_euler": xe, "logp_euler": le["log_p"], "x_nocond": xn, "logp_nocond": ln["log_p"]}
    return run


SCENARIOS["edm_classifier_guidance"] = edm_classifier_guidance()


def _sample_fp64(agent, lib_kind: str, prior, zs, **kw):
    """`_sample` with the agent cast to float64 first (weights, tables and draws: the float32 values, cast up) -- the yardstick that
    tells how far the REFERENCE'S OWN fp32 result is from exact arithmetic on a scenario (gen_golden_extra.fp64_yardstick)."""
    agent.model.double()
    agent.model_ema.double()
    for holder in (agent.model, agent.model_ema):          # (the solvers build their timestep vectors in float32)
        net = holder["diffusion"]
        # (integer timesteps of the discrete solvers stay integers: the positional embedding truncates them)
        net.forward = (lambda f: lambda x, noise, condition=None: f(x, noise.double() if noise.is_floating_point() else noise, condition))(net.forward)
    for k, v in list(vars(agent).items()):
        if isinstance(v, torch.Tensor) and v.is_floating_point():
            setattr(agent, k, v.double())
    kw = {k: (v.double() if isinstance(v, torch.Tensor) and v.is_floating_point() else v) for k, v in kw.items()}
    torch.set_default_dtype(torch.float64)                  # (only around the call: the scenario's seeded draws are float32 draws)
    try:
        return _sample_fp32(agent, lib_kind, prior.double(), [z.double() for z in zs], **kw)
    finally:
        torch.set_default_dtype(torch.float32)


_sample_fp32 = _sample


_ROWS = None


def run(name: str, lib_kind: str, device="cpu", fp64: bool = False, rows=None):
    """Outputs of scenario `name` as {key: tensor}; keys starting with '_' are live objects for the caller, not results.
    `fp64`: the same scenario evaluated in float64 (CPU).  `rows` (config-4 scenarios): a slice of the batch -- the same draws, only
    those trajectories sampled."""
    global _sample, _ROWS
    torch.manual_seed(1234)
    _sample = _sample_fp64 if fp64 else _sample_fp32
    _ROWS = rows
    try:
        return (SCENARIOS.get(name) or GPU_ONLY[name])(cases.lib_namespace(lib_kind), lib_kind, device)
    finally:
        _sample, _ROWS = _sample_fp32, None
