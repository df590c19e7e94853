"""BUILD CONTAINER ONLY: let the REAL reference write a checkpoint with its own ``DiffusionModel.save()`` (reference
diffusion/basic.py:94-98) and sample from it, so that tests/ can check the compatibility contract end to end: a file written by
CleanDiffuser loads into this package (``agent.load``) and produces the reference's samples.  TEST INFRASTRUCTURE.

Writes tests/golden/ref_checkpoint_janner_tiny.pt (the reference's torch.save of {"model", "model_ema"} state dicts, ~120 KB) and
tests/golden/ref_checkpoint_janner_tiny.npz (samples the reference drew from that checkpoint on recorded noise).
Usage: python -m oracle.gen_reference_checkpoint"""
import os

import numpy as np
import torch

from . import cases

NAME = "janner_tiny_disc_ddpm"


def build(lib, device="cpu"):
    c = cases.CASES[NAME]
    torch.manual_seed(7)                               # the checkpoint carries the weights: plain torch init, not load_synth
    net = getattr(lib, c["net"][0])(**c["net"][1])
    fm = None
    inp = cases.make_inputs(NAME)
    if inp["fix_mask"] is not None:
        fm = torch.from_numpy(inp["fix_mask"])
    agent = getattr(lib, c["solver"][0])(net, None, fix_mask=fm, device=device, **c["solver"][1])
    return agent, inp


def main(out_dir="tests/golden"):
    lib = cases.lib_namespace("reference")
    agent, inp = build(lib)
    # make model and model_ema differ, as after training: one EMA step away from a perturbed model
    with torch.no_grad():
        for p in agent.model.parameters():
            p.add_(0.01 * torch.randn_like(p))
    agent.ema_update()
    path = os.path.join(out_dir, "ref_checkpoint_janner_tiny.pt")
    agent.save(path)
    agent.eval()
    kw = cases.sample_kwargs(NAME, inp)
    outs = {}
    for use_ema in (True, False):
        with cases.replay_randn(list(inp["noise"])):
            x, _ = agent.sample(torch.from_numpy(inp["prior"]), use_ema=use_ema, **kw)
        outs["x_ema" if use_ema else "x_model"] = x.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(out_dir, "ref_checkpoint_janner_tiny.npz"), **outs)
    print(path, os.path.getsize(path), {k: v.shape for k, v in outs.items()})


if __name__ == "__main__":
    main()
