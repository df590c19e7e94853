"""Generate tests/golden/*.npz by running the REAL reference (imported from /root/reference) on the cases of
oracle/cases.py.  Run in the build container:  ``python -m oracle.gen_golden``.  The fixtures are committed; the GPU
box never needs /root/reference.  TEST INFRASTRUCTURE -- see oracle/__init__.py.

Per case the fixture stores only small tensors: ``x_out`` (final samples), ``pred0`` (the first network
forward, for bisecting a mismatch into "backbone" vs "solver"), ``n_draws`` (Gaussian draws consumed).
Weights/inputs are NOT stored: they are re-derived from the PCG64 streams (cleandiffuser_amd/utils/synth.py).
"""
import os
import sys

import numpy as np
import torch

from . import cases


def run_reference_case(lib, name: str):
    torch.manual_seed(1234)
    agent, net = cases.build(lib, name)
    inp = cases.make_inputs(name)
    kw = cases.sample_kwargs(name, inp)
    prior = torch.from_numpy(inp["prior"])
    used = {"n": 0}

    def counting(noise):
        for z in noise:
            used["n"] += 1
            yield z

    with cases.replay_randn(counting(inp["noise"])):
        x, log = cases.sampler_of(agent, name)(prior, **kw)
    c = cases.CASES[name]
    if "x_shape" in c:                    # non-Janner cases: the loop output alone pins them
        out = dict(x_out=x.detach().numpy().astype(np.float32), n_draws=np.int64(used["n"]))
        if isinstance(log, dict) and log.get("log_p") is not None:
            out["log_p"] = log["log_p"].detach().numpy().astype(np.float32)
        if c["net"][0] in cases.BIGBATCH_NETS:   # stand-alone forward with per-sample timesteps
            fx, ft, fc = cases.forward_probe(name, agent, inp)
            with torch.no_grad():
                out["fwd_pred"] = agent.model_ema["diffusion"](fx, ft, fc).numpy().astype(np.float32)
        return out
    # first backbone forward on the initial state, at the first timestep the loop visits
    temp = c["sample"].get("temperature", 1.0)
    fm = torch.from_numpy(inp["fix_mask"])[None] if inp["fix_mask"] is not None else 0.
    xt0 = torch.from_numpy(inp["noise"][0]) * temp
    xt0 = xt0 * (1. - fm) + prior * fm
    S = c["sample"]["sample_steps"]
    if c["solver"][0] == "DiscreteDiffusionSDE":
        from cleandiffuser.utils import SUPPORTED_SAMPLING_STEP_SCHEDULE as SS
        sched = SS[c["sample"].get("sample_step_schedule", "uniform")](agent.diffusion_steps, S)
        t0 = torch.full((c["batch"],), int(sched[S]), dtype=torch.long)
    else:
        from cleandiffuser.utils import SUPPORTED_SAMPLING_STEP_SCHEDULE as SS
        sched = SS[c["sample"].get("sample_step_schedule", "uniform_continuous")](agent.t_diffusion, S)
        t0 = torch.full((c["batch"],), float(sched[S]), dtype=torch.float32)
    with torch.no_grad():
        cond = torch.from_numpy(inp["cond"]) if inp["cond"] is not None else None
        pred0 = agent.model_ema["diffusion"](xt0, t0, cond)
    out = dict(x_out=x.detach().numpy().astype(np.float32), pred0=pred0.numpy().astype(np.float32),
               n_draws=np.int64(used["n"]))
    if log.get("log_p") is not None:
        out["log_p"] = log["log_p"].detach().numpy().astype(np.float32)
    return out


def main(out_dir="tests/golden", only=None):
    lib = cases.lib_namespace("reference")
    os.makedirs(out_dir, exist_ok=True)
    for name in cases.CASES:
        if only and name not in only:
            continue
        out = run_reference_case(lib, name)
        assert np.isfinite(out["x_out"]).all(), name
        np.savez_compressed(os.path.join(out_dir, name.replace("+", "p") + ".npz"), **out)
        print(f"{name:40s} x_out{out['x_out'].shape} |x|max={np.abs(out['x_out']).max():.3f} "
              f"draws={int(out['n_draws'])}")


if __name__ == "__main__":
    main(only=set(sys.argv[1:]) or None)
