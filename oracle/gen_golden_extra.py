"""Generate tests/golden/extra_*.npz by running the REAL reference (imported from /root/reference) on oracle/extra_cases.py.
Build container only: ``python -m oracle.gen_golden_extra [name ...]``.  TEST INFRASTRUCTURE -- see oracle/__init__.py."""
import os
import sys

import numpy as np

from . import extra_cases


def dit_h40_depth8_fp64(out_dir="tests/golden"):
    """The depth-8 DiT1d scenario once more with the REFERENCE run in float64 (same fp32 weights and draws, cast up): on synthetic,
    saturating weights eight unnormalised residual blocks amplify fp32 rounding so much that the reference's own fp32 result is
    1.7e-4 away from this one -- the yardstick the GPU test uses for that single scenario next to the fp32 fixture."""
    import torch
    from cleandiffuser_amd.utils import load_synth
    from . import cases
    lib = cases.lib_namespace("reference")
    which, B, steps, x_shape = "dit_h40_depth8", 3, 3, (40, 29)
    net = lib.DiT1d(29, emb_dim=128, d_model=256, n_heads=8, depth=8, timestep_emb_type="fourier")
    agent = lib.DiscreteDiffusionSDE(load_synth(net, 31), lib.IdentityCondition(dropout=0.0), predict_noise=True,
                                     x_max=2 * torch.ones(1, *x_shape), x_min=-2 * torch.ones(1, *x_shape), diffusion_steps=20,
                                     device="cpu")
    agent.eval()
    g = torch.Generator().manual_seed(len(which))
    cond = torch.randn(B, 128, generator=g)
    zs = [torch.randn(B, *x_shape, generator=g) for _ in range(steps + 1)]
    torch.set_default_dtype(torch.float64)
    try:
        agent.model.double()
        agent.model_ema.double()
        for k, v in list(vars(agent).items()):
            if isinstance(v, torch.Tensor) and v.is_floating_point():
                setattr(agent, k, v.double())
        with cases.replay_randn([z.double() for z in zs]):
            x, _ = agent.sample(torch.zeros(B, *x_shape), solver="ddim", n_samples=B, sample_steps=steps, w_cfg=1.3,
                                condition_cfg=cond.double())
    finally:
        torch.set_default_dtype(torch.float32)
    x32 = np.load(os.path.join(out_dir, "extra_dit_h40_depth8.npz"))["x"]
    np.savez_compressed(os.path.join(out_dir, "extra_dit_h40_depth8_fp64.npz"), x=x.numpy())
    print(f"dit_h40_depth8 fp64: reference fp32 vs fp64 max|d| = {np.abs(x32 - x.numpy()).max():.3e}")


FP64_YARDSTICKS = ("baseline_cfg4_tied", "baseline_cfg4_tied_b96", "baseline_cfg4_tied_b512", "chitf_ta10", "dit_h96")


def fp64_yardstick(name, out_dir="tests/golden"):
    """Scenario `name` once more with the REFERENCE evaluated in float64 -> extra_<name>_fp64.npz.  Config 4 (eps-prediction, no clip,
    alpha(1) = 0.0066, CFG w = 2) amplifies fp32 rounding so much that the reference's own fp32 result is ~1.8e-4 away from this one
    (and ~3e-4 away from ITSELF on another CPU: EPYC vs Xeon BLAS paths): the GPU test measures the native path against this yardstick
    next to the fp32 fixture (tools/dit_error_budget.py has the full table)."""
    x64 = extra_cases.run(name, "reference", fp64=True)["x"].detach().cpu().numpy()[::extra_cases.SUBSAMPLED.get(name, 1)]
    x32 = np.load(os.path.join(out_dir, f"extra_{name}.npz"))["x"]
    np.savez_compressed(os.path.join(out_dir, f"extra_{name}_fp64.npz"), x=x64.astype(np.float32))       # (stored rounded: half the file)
    print(f"{name} fp64: reference fp32 vs fp64 max|d| = {np.abs(x32 - x64).max():.3e}  mean|d| = {np.abs(x32 - x64).mean():.3e}")


def main(out_dir="tests/golden", only=None):
    os.makedirs(out_dir, exist_ok=True)
    for name in list(extra_cases.SCENARIOS) + list(extra_cases.GPU_ONLY):
        if only and name not in only:
            continue
        rows = extra_cases.ROW_SUBSET.get(name)
        out = {k: v.detach().cpu().numpy().astype(np.float32) for k, v in extra_cases.run(name, "reference", rows=rows).items()
               if not k.startswith("_")}
        if rows is not None:
            out["rows"] = np.asarray(rows, np.int32)
        assert all(np.isfinite(v).all() for v in out.values()), name
        if name in extra_cases.SUBSAMPLED:               # every stride-th trajectory (+ the stride itself)
            st = extra_cases.SUBSAMPLED[name]
            out = {k: v[::st] for k, v in out.items()}
            out["stride"] = np.array([st], np.int32)
        np.savez_compressed(os.path.join(out_dir, f"extra_{name}.npz"), **out)
        print(f"{name:24s} " + " ".join(f"{k}{v.shape} |max|={np.abs(v).max():.3f}" for k, v in out.items()))


if __name__ == "__main__":
    main(only=sys.argv[1:] or None)
    if not sys.argv[1:] or "dit_h40_depth8" in sys.argv[1:]:
        dit_h40_depth8_fp64()
    for name in FP64_YARDSTICKS:
        if not sys.argv[1:] or name in sys.argv[1:]:
            fp64_yardstick(name)
