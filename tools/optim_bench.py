"""Roofline check of the optimiser kernels (cdx_optim_f32): the fused AdamW + EMA + zero-grad pass moves 40 B per parameter
(p, g, m, v, ema read; p, m, v, ema, g written), the norm pass 4 B -- HBM-bound, priced against the 8 TB/s of MI355X_MICROARCH.md.
Usage (GPU box): python tools/optim_bench.py   -> one JSON line per model size."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cleandiffuser_amd.engine.optim import FusedAdamW  # noqa: E402
from cleandiffuser_amd.nn_diffusion import ChiUNet1d, JannerUNet1d  # noqa: E402
from copy import deepcopy  # noqa: E402

DEV = torch.device("cuda", 0)
PEAK_GBS = 8000.0


def run(name, net):
    net = net.to(DEV)
    ema = deepcopy(net).requires_grad_(False)
    n = sum(p.numel() for p in net.parameters())
    for p in net.parameters():
        p.grad = torch.randn_like(p) * 0.01
    out = {"model": name, "parameters": n, "tensors": len(list(net.parameters()))}
    for label, mk, step in (
            ("fused", lambda: FusedAdamW(net.parameters(), lr=2e-4, weight_decay=1e-5),
             lambda o: o.step(max_norm=1.0, ema=(net, ema, 0.995), zero_grad=False)),
            ("torch", lambda: torch.optim.AdamW(net.parameters(), lr=2e-4, weight_decay=1e-5), None)):
        opt = mk()

        def torch_step(o):
            torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
            o.step()
            with torch.no_grad():
                for p, e in zip(net.parameters(), ema.parameters()):
                    e.mul_(0.995).add_(p.detach(), alpha=0.005)
        fn = step or torch_step
        for _ in range(3):
            fn(opt)
        torch.cuda.synchronize()
        reps = 20
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(opt)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[label + "_ms"] = 1e3 * dt
        if label == "fused":
            out["fused_gbs"] = 44.0 * n / dt / 1e9          # 40 B (AdamW + EMA pass) + 4 B (norm pass) per parameter
            out["fused_frac_of_hbm_peak"] = out["fused_gbs"] / PEAK_GBS
    out["speedup"] = out["torch_ms"] / out["fused_ms"]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    run("JannerUNet1d config 2 (3.96 M parameters)", JannerUNet1d(23, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5))
    run("ChiUNet1d config 3 (68.9 M parameters)", ChiUNet1d(2, 20, 2, model_dim=256, emb_dim=256, dim_mult=[1, 2, 2], obs_as_global_cond=True))
