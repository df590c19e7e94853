"""Secondary measurements: BASELINE configs 1 and 3 through the fused kernel (bench.py stays the config-2 contract).
Usage (GPU box): python tools/bench_configs.py [cfg1|cfg3 ...]   -> one JSON line per config."""
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


def cfg3():
    net = load_synth(ChiUNet1d(2, 20, 2, model_dim=256, emb_dim=256, dim_mult=[1, 2, 2], obs_as_global_cond=True))
    agent = DDPM(net, IdentityCondition(dropout=0.0), diffusion_steps=50, x_max=torch.ones(1, 16, 2, device=DEV),
                 x_min=-torch.ones(1, 16, 2, device=DEV), device=DEV)
    agent.eval()
    B = 1024
    cond = torch.randn(B, 2, 20, device=DEV)
    zs = [torch.randn(B, 16, 2, device=DEV) for _ in range(50)]
    call = lambda: agent.sample(torch.zeros(B, 16, 2, device=DEV), n_samples=B, sample_steps=50,  # noqa: E731
                                condition_cfg=cond, w_cfg=1.0, noise=zs)[0]
    return "config 3: ChiUNet1d dp_pusht H=16 act=2 obs=20, 50-step legacy DDPM, B=1024", call, B, 50, net, 16


from cleandiffuser_amd.engine.program import MLP_TILE as P_TILE  # noqa: E402


def run(name, fn, reps=3):
    label, call, B, steps, net, horizon = fn()
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
    prog = runtime.compiled_program(net, horizon).prog
    per_unit = horizon if prog.tile else 1                  # tile programs: MACs are per workgroup of `tile` samples
    flops = 2.0 * prog.macs_per_forward / per_unit * steps * B
    k = sum(k_ms) / len(k_ms)
    print(json.dumps({"config": label, "trajectories_per_s": B / dt, "ms_per_call": 1e3 * dt, "kernel_ms": k,
                      "launches_per_call": len(k_ms) / reps, "tflops": flops / (k * 1e-3) / 1e12,
                      "frac_fp32_mfma_peak": flops / (k * 1e-3) / 1e12 / PEAK,
                      "lds_bytes": prog.lds_floats * 4, "weights_mb": prog.blob.numel() * 4 / 1e6}), flush=True)


if __name__ == "__main__":
    for name in (sys.argv[1:] or ["cfg1", "cfg3"]):
        run(name, {"cfg1": cfg1, "cfg3": cfg3}[name])
