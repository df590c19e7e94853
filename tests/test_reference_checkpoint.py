"""Compatibility contract: a checkpoint written by the REFERENCE's ``DiffusionModel.save()`` (reference
diffusion/basic.py:94-98; file committed by oracle/gen_reference_checkpoint.py, build container) loads into this package with
``agent.load()`` -- same state_dict keys, same shapes -- and samples exactly what the reference sampled from it; and what this
package saves has the reference's layout."""
import os

import numpy as np
import pytest
import torch

from conftest import golden_path
from oracle import cases
from oracle.gen_reference_checkpoint import NAME, build

CKPT = os.path.join(os.path.dirname(golden_path("x")), "ref_checkpoint_janner_tiny.pt")


def _loaded(amd_lib, device):
    agent, inp = build(amd_lib, device)
    before = {k: v.clone() for k, v in agent.model_ema.state_dict().items()}
    agent.load(CKPT)
    agent.eval()
    after = agent.model_ema.state_dict()
    assert any(not torch.equal(before[k], after[k]) for k in before), "load() must replace the weights"
    return agent, inp


def test_reference_written_checkpoint_loads_and_reproduces_reference_samples(amd_lib, tmp_path):
    gold = np.load(golden_path("ref_checkpoint_janner_tiny"))
    ckpt = torch.load(CKPT, map_location="cpu")
    agent, inp = _loaded(amd_lib, "cpu")
    assert set(ckpt) == {"model", "model_ema"}
    assert list(ckpt["model"]) == list(agent.model.state_dict()), "state_dict keys / order must equal the reference's"
    kw = cases.sample_kwargs(NAME, inp)
    for use_ema, key in ((True, "x_ema"), (False, "x_model")):
        x, _ = agent.sample(torch.from_numpy(inp["prior"]), noise=list(inp["noise"]), use_ema=use_ema, **kw)
        np.testing.assert_allclose(x.numpy(), gold[key], rtol=2e-6, atol=2e-6)
    assert np.abs(gold["x_ema"] - gold["x_model"]).max() > 1e-4          # (the two weight sets really differ)
    # and back: what this package saves is the reference's layout, tensor for tensor
    out = str(tmp_path / "amd.pt")
    agent.save(out)
    mine = torch.load(out, map_location="cpu")
    assert set(mine) == set(ckpt)
    for part in ("model", "model_ema"):
        assert list(mine[part]) == list(ckpt[part])
        assert all(torch.equal(mine[part][k], ckpt[part][k]) for k in ckpt[part])


@pytest.mark.gpu
def test_reference_written_checkpoint_on_device(amd_lib):
    """Same file through the native path: load() replaces the weights in place, so the packed-program caches must notice."""
    gold = np.load(golden_path("ref_checkpoint_janner_tiny"))
    agent, inp = build(amd_lib, "cuda:0")
    agent.eval()
    kw = cases.sample_kwargs(NAME, inp, device="cuda:0")
    prior = torch.from_numpy(inp["prior"]).to("cuda:0")
    x0, _ = agent.sample(prior, noise=list(inp["noise"]), **kw)           # compiles + caches a program of the INITIAL weights
    agent.load(CKPT)
    for use_ema, key in ((True, "x_ema"), (False, "x_model")):
        x, _ = agent.sample(prior, noise=list(inp["noise"]), use_ema=use_ema, **kw)
        np.testing.assert_allclose(x.cpu().numpy(), gold[key], rtol=1e-4, atol=1e-4)
    assert not torch.allclose(x0.cpu(), torch.from_numpy(gold["x_ema"]), atol=1e-3)
