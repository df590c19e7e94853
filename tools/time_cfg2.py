"""A/B timing of BASELINE config 2 (JannerUNet1d, H = 32, D = 23, 20-step DDIM) under different launch modes -- tuning aid.
Usage (GPU box): python tools/time_cfg2.py B[:ENV=VAL[,ENV=VAL...]] ...   e.g.  256 256:CDX_UNET2_GROUP=0 256:CDX_UNET2_GROUP=2
Prints per setting: whole-call ms (back-to-back calls, one sync at the end) and the kernel's own duration (HIP events on its stream)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cleandiffuser_amd.engine import runtime, runtime2  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    agent, net = bench.build_agent(dev)
    for spec in sys.argv[1:]:
        b, _, envs = spec.partition(":")
        batch = int(b)
        sets = dict(e.split("=", 1) for e in envs.split(",") if e)
        old = {k: os.environ.get(k) for k in sets}
        os.environ.update(sets)
        try:
            prior, z0 = bench.make_inputs(dev, 0, batch)
            kw = dict(solver="ddim", n_samples=batch, sample_steps=20, temperature=0.5)
            for _ in range(5):
                x, _ = agent.sample(prior, **kw)
            torch.cuda.synchronize()
            reps = 100
            t0 = time.perf_counter()
            for _ in range(reps):
                x, _ = agent.sample(prior, **kw)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / reps
            runtime.enable_launch_timing(True)
            for _ in range(20):
                agent.sample(prior, **kw)
            k_ms = runtime.drain_launch_timing()
            r_ms = runtime.drain_repair_timing()
            runtime.enable_launch_timing(False)
            runtime2.check_split_errors()
            kern = sum(k_ms) / 20
            frac = batch * bench.FLOPS_PER_TRAJ / (kern * 1e-3) / (bench.PEAK_FP32_MFMA_TFLOPS * 1e12)
            print(f"B={batch} {envs or 'default':40s} {batch / ms * 1e3:9.0f} traj/s  ms_per_call {ms:.3f}  kernel_ms {kern:.3f}  "
                  f"fp32-MFMA frac {frac:.3f}  idle repair launch {1e3 * sum(r_ms) / max(len(r_ms), 1):.1f} us x{len(r_ms)}  finite={bool(torch.isfinite(x).all())}  modes ok: split {runtime2._split_ok.get(dev)} group {runtime2._group_ok.get(dev)}", flush=True)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v


if __name__ == "__main__":
    main()
