"""Secondary measurements: BASELINE configs 1 and 3 through the fused kernel, 4 and 5 (one GPU's shard) through the
big-batch executors (bench.py stays the config-2 contract).
Usage (GPU box): python tools/bench_configs.py [cfg1|cfg3|cfg4[:B]|cfg5[:B] ...]   -> one JSON line per config."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleandiffuser_amd.diffusion import DiscreteDiffusionSDE  # noqa: E402
from cleandiffuser_amd.diffusion.ddpm import DDPM  # noqa: E402
from cleandiffuser_amd.engine import runtime  # noqa: E402
from cleandiffuser_amd.nn_condition import IdentityCondition, PearceObsCondition  # noqa: E402
from cleandiffuser_amd.nn_diffusion import ChiUNet1d, PearceMlp  # noqa: E402
from cleandiffuser_amd.utils import load_synth  # noqa: E402

DEV = torch.device("cuda", 0)
PEAK = 157.3


def cfg1():
    net = load_synth(PearceMlp(6, To=1, emb_dim=64, hidden_dim=256))
    cond = load_synth(PearceObsCondition(17, 64, flatten=True, dropout=0.0), 2)
    agent = DiscreteDiffusionSDE(net, cond, predict_noise=False, x_max=torch.ones(1, 6), x_min=-torch.ones(1, 6),
                                 diffusion_steps=100, device=DEV)
    agent.eval()
    B = 256
    obs = torch.randn(B, 1, 17, device=DEV)
    zs = [torch.randn(B, 6, device=DEV) for _ in range(100)]
    call = lambda: agent.sample(torch.zeros(B, 6, device=DEV), solver="ddpm", n_samples=B, sample_steps=100,  # noqa: E731
                                temperature=0.5, w_cfg=1.0, condition_cfg=obs, noise=zs)[0]
    return "config 1: PearceMlp DDPM act=6 obs=17, 100 steps, B=256", call, B, 100, net, P_TILE


def cfg3(B=1024, dim=256):
    net = load_synth(ChiUNet1d(2, 20, 2, model_dim=dim, emb_dim=dim, dim_mult=[1, 2, 2], obs_as_global_cond=True))
    agent = DDPM(net, IdentityCondition(dropout=0.0), diffusion_steps=50, x_max=torch.ones(1, 16, 2, device=DEV),
                 x_min=-torch.ones(1, 16, 2, device=DEV), device=DEV)
    agent.eval()
    cond = torch.randn(B, 2, 20, device=DEV)
    zs = [torch.randn(B, 16, 2, device=DEV) for _ in range(50)]
    call = lambda: agent.sample(torch.zeros(B, 16, 2, device=DEV), n_samples=B, sample_steps=50,  # noqa: E731
                                condition_cfg=cond, w_cfg=1.0, noise=zs)[0]
    tag = "" if dim == 256 else f" (model_dim {dim})"
    return f"config 3: ChiUNet1d dp_pusht H=16 act=2 obs=20, 50-step legacy DDPM, B={B}{tag}", call, B, 50, net, 16


from cleandiffuser_amd.engine.consts import MLP_TILE as P_TILE  # noqa: E402  (cfg1's sixth return value: unused for tile programs)


def _time_calls(call, reps):
    x = call()
    torch.cuda.synchronize()
    assert torch.isfinite(x).all()
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def cfg4(B=512):
    """Config 4: DiT1d Decision Diffuser (d_model 320, 10 heads, depth 2, 64 tokens x 29), CFG w = 2 (doubled batch),
    10-step DPM-Solver++(2M).  BASELINE shards B = 4096 over 8 GPUs -> 512 trajectories per GPU."""
    from cleandiffuser_amd.diffusion import ContinuousDiffusionSDE
    from cleandiffuser_amd.nn_condition import MLPCondition
    from cleandiffuser_amd.nn_diffusion import DiT1d
    T, D, d, depth, steps = 64, 29, 320, 2, 10
    net = load_synth(DiT1d(D, emb_dim=128, d_model=d, n_heads=10, depth=depth, timestep_emb_type="fourier"))
    cond = load_synth(MLPCondition(1, 128, [128], torch.nn.SiLU(), dropout=0.25), 2)
    fix = torch.zeros(T, D)
    fix[0] = 1.0
    agent = ContinuousDiffusionSDE(net, cond, fix_mask=fix, predict_noise=True, noise_schedule="linear",
                                   x_max=3 * torch.ones(1, T, D), x_min=-3 * torch.ones(1, T, D), device=DEV)
    agent.eval()
    prior = torch.zeros(B, T, D, device=DEV)
    prior[:, 0] = torch.randn(B, D, device=DEV)
    ret = torch.rand(B, 1, device=DEV)
    z = [torch.randn(B, T, D, device=DEV)]
    call = lambda: agent.sample(prior, solver="ode_dpmsolver++_2M", n_samples=B, sample_steps=steps, w_cfg=2.0,  # noqa: E731
                                temperature=0.5, condition_cfg=ret, noise=z)[0]
    tok_macs = D * d + depth * (3 * d * d + d * d + 8 * d * d) + depth * 2 * T * d + d * D   # per token, attention incl.
    smp_macs = 128 * d + d * d + depth * 6 * d * d + 2 * d * d                              # per (doubled) sample
    flops = 2.0 * (tok_macs * T + smp_macs) * 2 * steps * B
    return f"config 4 shard: DiT1d d=320 h=10 depth=2, H=64 D=29, CFG w=2, 10-step dpmsolver++2M, B={B}", call, B, flops


def cfg5(B=131072, steps=128, D=15):
    """Config 5: SynthER residual MLP (IDQLMlp hidden 1024 x 6 blocks, emb 128, D = 15), 128-step EDM Euler.
    BASELINE shards B = 1 M over 8 GPUs -> 125 000 samples per GPU (the executor cuts a request into 16 384-row chunks itself).
    D = 27: the real hopper transition width 2 * 11 + 3 + 2 (report both)."""
    from cleandiffuser_amd.diffusion import ContinuousEDM
    from cleandiffuser_amd.nn_diffusion import IDQLMlp
    H, nb = 1024, 6
    net = load_synth(IDQLMlp(0, D, emb_dim=128, hidden_dim=H, n_blocks=nb))
    agent = ContinuousEDM(net, None, device=DEV)
    agent.eval()
    z = [torch.randn(B, D, device=DEV)]
    call = lambda: agent.sample(torch.zeros(B, D, device=DEV), solver="euler", n_samples=B, sample_steps=steps,  # noqa: E731
                                noise=z)[0]
    macs = (D + 128) * H + nb * 8 * H * H + H * D
    return f"config 5 shard: IDQLMlp 1024x6, D={D}, {steps}-step EDM Euler, B={B}", call, B, 2.0 * macs * steps * B


def cfg2big(B=3200):
    """Config 2's network at the real Diffuser batch (50 environments x 64 candidate plans), unguided 20-step DDIM."""
    from cleandiffuser_amd.nn_diffusion import JannerUNet1d
    H, D = 32, 23
    net = load_synth(JannerUNet1d(D, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5))
    fix = torch.zeros(H, D)
    fix[0, :17] = 1.0
    agent = DiscreteDiffusionSDE(net, None, fix_mask=fix, diffusion_steps=20, predict_noise=False, device=DEV)
    agent.eval()
    prior = torch.zeros(B, H, D, device=DEV)
    prior[:, 0, :17] = torch.randn(B, 17, device=DEV)
    z = [torch.randn(B, H, D, device=DEV)]
    call = lambda: agent.sample(prior, solver="ddim", n_samples=B, sample_steps=20, temperature=0.5, noise=z)[0]  # noqa: E731
    return f"config 2 network, 20-step DDIM, B={B}", call, B, 2.0 * 19.67e6 * 20 * B


def cfgT(B=1024, steps=100):
    """Diffusion Policy's transformer (dp_pusht): ChiTransformer d_model 256, 4 heads, 8 decoder layers, Ta=16, To=2, obs 20;
    100-step DDPM over DiscreteDiffusionSDE, conditional (w_cfg = 1)."""
    from cleandiffuser_amd.nn_diffusion import ChiTransformer
    Ta, A, d, L = 16, 2, 256, 8
    net = load_synth(ChiTransformer(A, 20, Ta, 2, d_model=d, nhead=4, num_layers=L))
    one = torch.ones(1, Ta, A)
    agent = DiscreteDiffusionSDE(net, IdentityCondition(dropout=0.0), predict_noise=True, x_max=one, x_min=-one,
                                 diffusion_steps=steps, device=DEV)
    agent.eval()
    obs = torch.randn(B, 2, 20, device=DEV)
    zs = [torch.randn(B, Ta, A, device=DEV) for _ in range(steps)]
    call = lambda: agent.sample(torch.zeros(B, Ta, A, device=DEV), solver="ddpm", n_samples=B, sample_steps=steps,  # noqa: E731
                                condition_cfg=obs, w_cfg=1.0, noise=zs)[0]
    tok = A * d + L * (3 * d * d + d * d + d * d + d * d + 8 * d * d) + L * (2 * Ta * d + 2 * 3 * d) + d * A     # MACs per token
    return f"ChiTransformer dp_pusht d=256 h=4 L=8, Ta=16, {steps}-step DDPM, B={B}", call, B, 2.0 * tok * Ta * steps * B


def cfgU(B=256, native_backward=None, graph=False):
    """update() of config 2 (row f4): one training step on a batch of B trajectories -- forward + backward of the denoising loss,
    gradient-norm clip, AdamW, EMA.  Returns (label, call, B): `call` runs ONE update and returns the loss tensor."""
    from cleandiffuser_amd.nn_diffusion import JannerUNet1d
    H, D = 32, 23
    net = load_synth(JannerUNet1d(D, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5))
    fix = torch.zeros(H, D)
    fix[0, :17] = 1.0
    agent = DiscreteDiffusionSDE(net, None, fix_mask=fix, diffusion_steps=20, predict_noise=False, grad_clip_norm=1.0, device=DEV)
    x0 = torch.randn(B, H, D, device=DEV)
    if native_backward is not None:
        os.environ["CDX_TRAIN_NATIVE"] = "1" if native_backward else "0"
    os.environ["CDX_TRAIN_GRAPH"] = "auto" if graph else "0"
    call = lambda: torch.as_tensor(agent.update(x0)["loss"])  # noqa: E731
    return f"config 2 update(): JannerUNet1d H=32 D=23, batch {B}, loss + backward + clip + AdamW + EMA", call, B


def cfgUC(B=256, native_backward=None, graph=False):
    """One Diffuser TRAINING ITERATION at config-2 size (reference pipelines/diffuser_d4rl_mujoco.py:88-91): ``agent.update(x0)`` of the
    denoiser next to ``agent.update_classifier(x0, R)`` of the CumRewClassifier(HalfJannerUNet1d) on the same batch.  Returns
    (label, call, B, macs): `macs` = multiply-adds of ONE forward of denoiser + classifier per trajectory (a step is ~3x that)."""
    from cleandiffuser_amd.classifier import CumRewClassifier
    from cleandiffuser_amd.engine import program2
    from cleandiffuser_amd.nn_classifier import HalfJannerUNet1d
    from cleandiffuser_amd.nn_diffusion import JannerUNet1d
    H, D = 32, 23
    net = load_synth(JannerUNet1d(D, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5))
    cnet = load_synth(HalfJannerUNet1d(H, D, out_dim=1, kernel_size=3, model_dim=32, emb_dim=32, dim_mult=(1, 2, 2, 2)), 1)
    if native_backward is not None:
        os.environ["CDX_TRAIN_NATIVE"] = "1" if native_backward else "0"
    os.environ["CDX_TRAIN_GRAPH"] = "auto" if graph else "0"
    macs = program2.compile_janner2(net, H, nw=8).macs_per_forward + program2.compile_classifier2(cnet, H).macs_per_forward
    fix = torch.zeros(H, D)
    fix[0, :17] = 1.0
    clf = CumRewClassifier(cnet, device=DEV)
    if native_backward is False:                           # the stock sequence: autograd over ATen kernels, torch.optim.Adam
        clf.optim = torch.optim.Adam(clf.model.parameters(), lr=2e-4, weight_decay=1e-4)
    agent = DiscreteDiffusionSDE(net, None, fix_mask=fix, classifier=clf, diffusion_steps=20, predict_noise=False, grad_clip_norm=1.0, device=DEV)
    x0, ret = torch.randn(B, H, D, device=DEV), torch.randn(B, 1, device=DEV)

    def call():
        loss = agent.update(x0)["loss"]
        agent.update_classifier(x0, ret)
        return torch.as_tensor(loss)
    return f"Diffuser training iteration at config-2 size: update() + update_classifier(), batch {B}", call, B, macs


def cfgD(B=64, steps=200_000, resident=True):
    """Batch supply of the Diffuser training loop (row f4, third slice): a hopper-sized synthetic D4RL dictionary (o = 11, a = 3), H = 32
    windows, batch B.  resident: D4RLMuJoCoDataset.loader (buffers in HBM, one gather launch per batch); else the reference's way --
    torch DataLoader(shuffle, drop_last) over the same dataset + .to(device) per field (num_workers = 0 here: worker processes only
    hide part of the same host work).  Returns (label, next_batch, B, resident_bytes)."""
    import numpy as np
    from cleandiffuser_amd.dataset.d4rl_mujoco_dataset import D4RLMuJoCoDataset
    from cleandiffuser_amd.utils import loop_dataloader
    rng = np.random.default_rng(0)                             # a D4RL-shaped dictionary: episodes of 1-1000 steps
    ends = np.minimum(np.cumsum(rng.integers(1, 1001, size=steps // 400 + 8)), steps) - 1
    term = np.zeros(steps, dtype=bool)
    term[ends] = True
    data = dict(observations=rng.standard_normal((steps, 11)).astype(np.float32), actions=rng.uniform(-1, 1, (steps, 3)).astype(np.float32),
                rewards=rng.standard_normal(steps).astype(np.float32), terminals=term, timeouts=np.zeros(steps, dtype=bool))
    ds = D4RLMuJoCoDataset(data, horizon=32, max_path_length=1000)
    if resident:
        it = loop_dataloader(ds.loader(B, device=DEV))
        nxt = lambda: next(it)  # noqa: E731
    else:
        from torch.utils.data import DataLoader
        it = loop_dataloader(DataLoader(ds, batch_size=B, shuffle=True, drop_last=True))

        def nxt():
            b = next(it)
            return {"obs": {"state": b["obs"]["state"].to(DEV)}, "act": b["act"].to(DEV), "rew": b["rew"].to(DEV), "val": b["val"].to(DEV)}
    how = "HBM-resident buffers, one cdx_gather_windows_f32 launch per batch" if resident else "torch DataLoader + H2D copy per field"
    return f"D4RL-MuJoCo sequence batches ({steps} steps, o=11 a=3, H=32), batch {B}: {how}", nxt, B, ds.resident_bytes()


def _guided_macs(net, clf_net, horizon, fallback):
    """MACs of ONE guided step per trajectory -- denoiser forward + classifier forward + classifier backward-data -- as the guided
    program compiler counts them from the packed ops (the guided roofline fractions used to price the
    denoiser's FLOPs only); `fallback` (denoiser only) when no guided program exists for the net."""
    from cleandiffuser_amd.engine import runtime2
    comp = runtime2.compiled_guided2(net.to(DEV), clf_net.to(DEV), horizon)
    return float(comp.prog.macs_per_forward) if comp.prog is not None else float(fallback)


def cfg2g(B=256):
    """Config 2 with classifier guidance at every step (w_cg > 0, what the shipped Diffuser configurations run): JannerUNet1d
    denoiser (fused forward per step) + CumRewClassifier(HalfJannerUNet1d) gradient per step, 20-step DDPM."""
    from cleandiffuser_amd.classifier import CumRewClassifier
    from cleandiffuser_amd.nn_classifier import HalfJannerUNet1d
    from cleandiffuser_amd.nn_diffusion import JannerUNet1d
    H, D = 32, 23
    net = load_synth(JannerUNet1d(D, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5))
    clf_net = load_synth(HalfJannerUNet1d(H, D, out_dim=1, model_dim=32, emb_dim=32, dim_mult=(1, 2, 2, 2), kernel_size=3), 1)
    fix = torch.zeros(H, D)
    fix[0, :17] = 1.0
    agent = DiscreteDiffusionSDE(net, None, fix_mask=fix, classifier=CumRewClassifier(clf_net, device=DEV), diffusion_steps=20,
                                 predict_noise=False, device=DEV)
    agent.eval()
    prior = torch.zeros(B, H, D, device=DEV)
    prior[:, 0, :17] = torch.randn(B, 17, device=DEV)
    cg = torch.ones(B, 1, device=DEV)
    call = lambda: agent.sample(prior, solver="ddpm", n_samples=B, sample_steps=20, temperature=0.5, w_cg=0.1,  # noqa: E731
                                condition_cg=cg)[0]
    flops = 2.0 * _guided_macs(net, clf_net, H, 19.67e6) * 20 * B      # denoiser + classifier forward + backward, 20 steps
    return f"config 2 + classifier guidance (w_cg=0.1, HalfJannerUNet1d), 20-step DDPM, B={B}", call, B, flops


def _shipped_diffuser(size, B, guided):
    """The two shipped Diffuser sizes with model_dim 64 (reference configs/diffuser/kitchen/kitchen.yaml: H = 32, D = 69;
    configs/diffuser/antmaze/antmaze.yaml: H = 64, D = 37), HalfJannerUNet1d classifier, 20-step DDPM, with / without guidance."""
    from cleandiffuser_amd.classifier import CumRewClassifier
    from cleandiffuser_amd.nn_classifier import HalfJannerUNet1d
    from cleandiffuser_amd.nn_diffusion import JannerUNet1d
    H, D, n_obs = (32, 69, 60) if size == "kitchen" else (64, 37, 29)
    net = load_synth(JannerUNet1d(D, model_dim=64, emb_dim=64, dim_mult=[1, 2, 2, 2], kernel_size=5))
    clf_net = load_synth(HalfJannerUNet1d(H, D, out_dim=1, model_dim=64, emb_dim=64, dim_mult=(1, 2, 2, 2), kernel_size=3), 1)
    fix = torch.zeros(H, D)
    fix[0, :n_obs] = 1.0
    agent = DiscreteDiffusionSDE(net, None, fix_mask=fix, classifier=CumRewClassifier(clf_net, device=DEV), diffusion_steps=20,
                                 predict_noise=False, device=DEV)
    agent.eval()
    prior = torch.zeros(B, H, D, device=DEV)
    prior[:, 0, :n_obs] = torch.randn(B, n_obs, device=DEV)
    call = lambda: agent.sample(prior, solver="ddpm", n_samples=B, sample_steps=20, temperature=0.5,  # noqa: E731
                                w_cg=0.1 if guided else 0.0)[0]
    macs = 4 * 19.67e6 * H / 32             # denoiser only, ~4x the config-2 net per position
    if guided:
        macs = _guided_macs(net, clf_net, H, macs)
    return (f"{size}-size Diffuser (model_dim 64, H={H}, D={D}), {'classifier guidance' if guided else 'unguided'}, 20-step DDPM, "
            f"B={B}"), call, B, 2.0 * macs * 20 * B


def cfgKg(B=256):
    return _shipped_diffuser("kitchen", B, True)


def cfgAg(B=256):
    return _shipped_diffuser("antmaze", B, True)


def cfgAu(B=256):
    return _shipped_diffuser("antmaze", B, False)


def run_big(name, fn, reps=2, **kw):
    label, call, B, flops = fn(**kw)
    dt = _time_calls(call, reps)
    print(json.dumps({"config": label, "samples_per_s": B / dt, "ms_per_call": 1e3 * dt, "tflops": flops / dt / 1e12,
                      "frac_fp32_mfma_peak": flops / dt / 1e12 / PEAK}), flush=True)


def run(name, fn, reps=3, **kw):
    label, call, B, steps, net, horizon = fn(**kw)
    x = call()
    torch.cuda.synchronize()
    assert torch.isfinite(x).all()
    runtime.enable_launch_timing(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    k_ms = runtime.drain_launch_timing()
    runtime.enable_launch_timing(False)
    from cleandiffuser_amd.engine import runtime2
    if runtime._mlp_kind(net) is not None:                  # tile programs: MACs are per workgroup of `tile` samples
        per_unit = runtime.mlp_tile(B)
        prog = runtime2.compiled_mlp2(net, runtime._mlp_kind(net), per_unit).prog
    else:
        per_unit, prog = 1, runtime2.shape_for(net, horizon, B)[0].prog
    flops = 2.0 * prog.macs_per_forward / per_unit * steps * B
    if not k_ms:                                            # served by the implicit-GEMM executor: no single fused launch to time
        print(json.dumps({"config": label + " [implicit-GEMM executor]", "trajectories_per_s": B / dt, "ms_per_call": 1e3 * dt,
                          "tflops": flops / dt / 1e12, "frac_fp32_mfma_peak": flops / dt / 1e12 / PEAK}), flush=True)
        return
    k = sum(k_ms) / len(k_ms)
    print(json.dumps({"config": label, "trajectories_per_s": B / dt, "ms_per_call": 1e3 * dt, "kernel_ms": k,
                      "launches_per_call": len(k_ms) / reps, "tflops": flops / (k * 1e-3) / 1e12,
                      "frac_fp32_mfma_peak": flops / (k * 1e-3) / 1e12 / PEAK,
                      "lds_bytes": prog.lds_bytes(1), "weights_mb": prog.blob.numel() * 4 / 1e6}), flush=True)


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["cfg1", "cfg3"]):
        if name.startswith(("cfg4", "cfg5", "cfg2g", "cfg2big", "cfgT", "cfgKg", "cfgAg", "cfgAu")):    # e.g. cfg4, cfg4:4096, cfg2g:3200, cfgT:256
            base, _, b = name.partition(":")                        # cfgT:1024:10 = batch 1024, 10 denoising steps (short profiles)
            b, _, st = b.partition(":")
            run_big(name, {"cfg4": cfg4, "cfg5": cfg5, "cfg2g": cfg2g, "cfg2big": cfg2big, "cfgT": cfgT, "cfgKg": cfgKg, "cfgAg": cfgAg, "cfgAu": cfgAu}[base],
                    **({"B": int(b)} if b else {}), **({"steps": int(st)} if st else {}))
        else:
            base, _, b = name.partition(":")                        # cfg3:64:32 = batch 64, model_dim 32
            b, _, dim = b.partition(":")
            run(name, {"cfg1": cfg1, "cfg3": cfg3}[base], **({"B": int(b)} if b else {}), **({"dim": int(dim)} if dim else {}))
