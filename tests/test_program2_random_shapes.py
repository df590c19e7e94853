"""Program compiler over shapes nobody picked by hand: seeded random nets through `engine/program2.py` and the kernel's CPU twin
(`oracle/lane_sim2.py`), against the modules' own forward (bit-identical to the reference's, tests/test_module_mirrors.py) and, for
the guided programs, torch.autograd.  A net the compiler cannot place must say so with a ValueError (the runtime's "not mine" signal:
the request then goes to the GEMM executor) -- any other exception, or a forward that differs, is a bug.

The sweep this file was cut from ran a few hundred nets per family; what it found is pinned below by name: slots narrower than their
16-row tile (8 and 24 channels: pad rows that no epilogue stores), dim_mult[0] != 1 (the reference's own forward fails on it), a
classifier so narrow that GroupNorm1d ends up with zero groups.
"""
import random

import numpy as np
import pytest
import torch

from cleandiffuser_amd.engine import program2 as P2
from oracle.lane_sim2 import LaneSim2, emb_table, run_forward_split

TOL = 5e-5


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1.0, float(np.abs(b).max())))


def _janner(amd_lib, D, md, dm, ks, seed, emb=None):
    from cleandiffuser_amd.utils import load_synth
    return load_synth(amd_lib.JannerUNet1d(D, model_dim=md, emb_dim=emb or md, dim_mult=dm, kernel_size=ks), seed).eval()


def _forward_case(net, H, seed):
    g = torch.Generator().manual_seed(seed)
    x, t = torch.randn(1, H, net.in_dim if hasattr(net, "in_dim") else net.final_conv[-1].out_channels, generator=g), torch.tensor([seed % 100])
    with torch.no_grad():
        return x, t, net._forward_torch(x, t, None)[0].numpy(), net.map_noise(t).numpy()


def _draw(rng):
    md = rng.choice([8, 16, 32, 64])
    dm = rng.choice([[1], [1, 2], [1, 2, 2], [1, 2, 4], [1, 1], [1, 3], [1, 4], [2, 1], [1, 2, 2, 2]])
    H = rng.choice([4, 8, 16, 32, 64])
    return md, dm, H, rng.randint(1, 40), rng.choice([3, 5, 7])


@pytest.mark.parametrize("seed", range(6))
def test_random_janner_nets_compile_to_the_module_forward_or_refuse(seed, amd_lib):
    rng = random.Random(1000 + seed)
    done = 0
    while done < 2:
        md, dm, H, D, ks = _draw(rng)
        if H % (2 ** (len(dm) - 1)):
            continue
        net = _janner(amd_lib, D, md, dm, ks, seed)
        try:
            prog = P2.compile_janner2(net, H, nw=rng.choice([4, 8]), compact=rng.random() < 0.3)
        except ValueError:
            continue
        assert prog.lds_bytes(1) <= 160 * 1024
        x, t, want, temb = _forward_case(net, H, seed)
        sim = LaneSim2(prog)
        for _ in range(2 if prog.compact else 1):                # (compact programs: the state slot is arena memory, reused)
            sim.poison_arena()
            sim.load_x(x[0].numpy())
            assert _rel(sim.run_forward(emb_table(prog, temb)[0]), want) < TOL, (md, dm, H, D, ks)
        done += 1


@pytest.mark.parametrize("shape", [(8, [1], 8, 37, 3), (8, [1, 4], 4, 18, 5), (8, [1, 3], 8, 17, 3), (8, [1, 2, 2, 2], 16, 3, 5)])
def test_slots_narrower_than_a_row_tile(shape, amd_lib):
    """8 and 24 channels (model_dim 8, x3): GroupNorm groups of 4 fit the epilogue partition, but the slot's 16-row tile has rows no
    epilogue ever stores.  The kernel clears its LDS once per launch and the consumers' weights are zero on those rows; the twin
    poisons LDS with NaN and accepts such a read only under all-zero weights."""
    md, dm, H, D, ks = shape
    net = _janner(amd_lib, D, md, dm, ks, 3, emb=32)
    prog = P2.compile_janner2(net, H, nw=4)
    x, t, want, temb = _forward_case(net, H, 5)
    sim = LaneSim2(prog)
    sim.load_x(x[0].numpy())
    assert _rel(sim.run_forward(emb_table(prog, temb)[0]), want) < TOL


def test_nets_the_reference_cannot_run_are_refused_not_asserted(amd_lib):
    from cleandiffuser_amd.engine import runtime2
    from cleandiffuser_amd.utils import load_synth
    net = _janner(amd_lib, 5, 16, [2, 1], 5, 0)               # final_conv is built on model_dim, the up path ends on 2 x model_dim
    with pytest.raises(ValueError, match="final conv"):
        P2.compile_janner2(net, 32)
    assert runtime2.supported(net, 32) is not None
    den = _janner(amd_lib, 38, 32, [1, 2, 2], 5, 0)
    clf = load_synth(amd_lib.HalfJannerUNet1d(64, 38, out_dim=1, model_dim=8, emb_dim=8, dim_mult=(1, 1), kernel_size=3), 1).eval()
    with pytest.raises(ValueError):                            # 2-channel layers: GroupNorm1d(min 4 channels per group) has no group
        P2.compile_guided2(den, clf, 64, save_global=True)
    assert runtime2.guided_supported(den, clf, 64) is not None


@pytest.mark.parametrize("seed", range(4))
def test_random_split_programs_agree_with_the_module(seed, amd_lib, monkeypatch):
    rng = random.Random(2000 + seed)
    monkeypatch.setattr(P2, "SPLIT_MIN_RECORDS", 0)
    while True:
        md, dm, H, D, ks = _draw(rng)
        if H % (2 ** (len(dm) - 1)) or md < 32:
            continue
        net, k = _janner(amd_lib, D, md, dm, ks, seed), rng.choice([2, 4])
        try:
            prog = P2.compile_janner2_split(net, H, k)
        except ValueError:
            continue
        x, t, want, temb = _forward_case(net, H, seed)
        sims = [LaneSim2(prog, member=m) for m in range(k)]
        for s in sims:
            s.load_x(x[0].numpy())
        assert _rel(run_forward_split(sims, emb_table(prog, temb)[0]), want) < TOL, (md, dm, H, D, ks, k)
        return


@pytest.mark.parametrize("seed", range(4))
def test_random_guided_programs_match_autograd(seed, amd_lib):
    from cleandiffuser_amd.utils import load_synth
    rng = random.Random(3000 + seed)
    variants = [{}, dict(save_global=True), dict(save_global=True, max_stage=2304), dict(save_global=True, compact=True),
                dict(save_global=True, compact=True, max_stage=2304)]
    while True:
        md, dm, H, D, ks = _draw(rng)
        cmd, cdm, cks = rng.choice([16, 32, 64]), rng.choice([dm, [1, 2], [1, 1]]), rng.choice([3, 5])
        if H % (2 ** (len(dm) - 1)) or H % (2 ** len(cdm)) or H > 32:
            continue
        net = _janner(amd_lib, D, md, dm, ks, seed)
        clf = load_synth(amd_lib.HalfJannerUNet1d(H, D, out_dim=1, model_dim=cmd, emb_dim=cmd, dim_mult=tuple(cdm), kernel_size=cks), seed + 1).eval()
        try:
            prog = P2.compile_guided2(net, clf, H, **rng.choice(variants))
        except ValueError:
            continue
        x, t, want, temb = _forward_case(net, H, seed)
        xr = x.clone().requires_grad_()
        clf._forward_torch(xr, t, None).sum().backward()
        grad = xr.grad[0].numpy()
        with torch.no_grad():
            row = emb_table(prog, None, [temb, clf.map_noise(t).numpy()])[0]
        sim = LaneSim2(prog)
        sim.poison_arena()
        sim.load_x(x[0].numpy())
        assert _rel(sim.run_forward(row), want) < TOL, (md, dm, H, D, ks, cmd, cdm, cks)
        assert float(np.abs(sim.grad() - grad).max()) < 2e-4 * max(1e-3, float(np.abs(grad).max())), (md, dm, H, D, ks, cmd, cdm, cks)
        return
