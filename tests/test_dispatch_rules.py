"""Host-side routing rules between the two native U-Net executors and the PyTorch executor (engine/dispatch.py, bigbatch.py,
runtime.py) -- pure Python decisions, checked on the CPU."""
import os
import torch

from cleandiffuser_amd.engine import bigbatch, plan as P, runtime, runtime2
from cleandiffuser_amd.nn_diffusion import ChiUNet1d, DiT1d, JannerUNet1d


def test_unet_executor_choice(monkeypatch):
    janner = JannerUNet1d(6, model_dim=16, emb_dim=16, dim_mult=[1, 2], kernel_size=5)
    chi = ChiUNet1d(2, 5, 1, model_dim=32, emb_dim=32, dim_mult=[1, 2], obs_as_global_cond=True)
    chi_local = ChiUNet1d(2, 5, 1, model_dim=32, emb_dim=32, dim_mult=[1, 2], obs_as_global_cond=False)
    assert not bigbatch.is_chiunet_gemm(janner, 256) and bigbatch.is_chiunet_gemm(janner, bigbatch.JANNER_GEMM_MIN_BATCH)
    assert not bigbatch.is_chiunet_gemm(chi, 8) and bigbatch.is_chiunet_gemm(chi, bigbatch.UNET_GEMM_MIN_BATCH)
    # a large net goes to the GEMM executor at every batch: the program kernel would re-stream its weights per trajectory
    monkeypatch.setattr(bigbatch, "UNET_GEMM_MIN_PARAMS", sum(p.numel() for p in chi.parameters()))
    assert bigbatch.is_chiunet_gemm(chi, 8)
    monkeypatch.setattr(bigbatch, "UNET_GEMM_MIN_PARAMS", 10 ** 7)
    assert not bigbatch.is_chiunet_gemm(chi, 8)
    # local conditioning: the implicit-GEMM executor is its only native path, at every batch size
    assert bigbatch.is_chiunet_gemm(chi_local, 1) and bigbatch.is_chiunet_gemm(chi_local, 10 ** 6)
    assert not bigbatch.is_chiunet_gemm(DiT1d(4, 8, d_model=16, n_heads=2, depth=1), 10 ** 6)
    # the program kernel's answer decides below the crossover batch (faked here: nothing with a horizon >= 64 fits)
    monkeypatch.setattr(runtime2, "supported", lambda module, horizon: "LDS plan needs 200000 B" if horizon >= 64 else None)
    monkeypatch.setattr(runtime2, "compact_only", lambda module, horizon: False)
    assert not bigbatch.is_chiunet_gemm(janner, 3, 32)                  # fits the program kernel: small batches stay there
    assert bigbatch.is_chiunet_gemm(janner, 3, 64)                      # does not fit: GEMM executor at any batch
    assert not bigbatch.is_chiunet_gemm(chi, 3, 16, True)               # EDM state lives in the launch workspace: no extra LDS
    assert bigbatch.is_chiunet_gemm(chi, 3, 64, True)                   # no program holds it: GEMM executor
    assert not bigbatch.is_chiunet_gemm(janner, 5000, 32)               # the program kernel keeps every batch size it can run
    assert bigbatch.is_chiunet_gemm(janner, 5000, 32, True) and bigbatch.is_chiunet_gemm(janner, 5000, 64)   # large EDM batches / too big: GEMM
    # nets that fit only as a compact program: sampling loops stay on the kernel, stand-alone forwards take the executor
    monkeypatch.setattr(runtime2, "compact_only", lambda module, horizon: True)
    assert not bigbatch.is_chiunet_gemm(janner, 3, 32) and bigbatch.is_chiunet_gemm(janner, 3, 32, forward=True)


def test_edm_plans_do_not_change_the_lds_plan():
    """EDM / consistency step kinds keep their state (slope, x_old) in the launch's global workspace: the same program serves EDM
    and non-EDM plans -- the shipped Diffuser kitchen net fits one workgroup either way."""
    euler = P.SamplePlan(solver="x", steps=[P.Step(P.KIND_EDM_EULER, P.V_EPS, 1, 0.1, 1.0, 1.0, (1.0, 1.0, 1.0, 0.1, 0.0))])
    ddim = P.SamplePlan(solver="ddim", steps=[P.Step(P.KIND_DDIM, P.V_EPS, 1, 3, 0.9, 0.4, (1.0, 0.4, 0.9, 0.0, 0.0))])
    assert runtime.plan_is_edm(euler) and not runtime.plan_is_edm(ddim)
    kitchen = JannerUNet1d(69, model_dim=64, emb_dim=64, dim_mult=[1, 2, 2, 2], kernel_size=5)
    assert runtime2.supported(kitchen, 32) is None and runtime.supported_backbone(kitchen, 32, True) is None



def test_launch_plan_cuts_large_batches():
    """B = 3200 = 3 rounds of 768 trajectories three per workgroup + 2 rounds of two per workgroup (13 x 256 places; the older rule's
    4 rounds of three + a half-empty round of one cost 2 % more on the MI355X); B = 512 one round of two; B = 256 one of one."""
    from cleandiffuser_amd.engine import runtime2
    assert runtime2.plan_parts(3200, 3) == [(0, 2304, 3), (2304, 896, 2)]
    assert runtime2.plan_parts(3072, 3) == [(0, 3072, 3)] and runtime2.plan_parts(768 * 40, 3) == [(0, 768 * 40, 3)]
    assert runtime2.plan_parts(10 ** 6, 3) == [(0, 999168, 3), (999168, 832, 2)]
    assert runtime2.plan_parts(512, 3) == [(0, 512, 2)]
    assert runtime2.plan_parts(256, 3) == [(0, 256, 1)]
    assert runtime2.plan_parts(640, 3) == [(0, 640, 3)]
    assert runtime2.plan_parts(3200, 2) == [(0, 3072, 2), (3072, 128, 1)]
    cost = lambda parts: sum(-(-c // (256 * t)) * runtime2.ROUND_COST[t] for _, c, t in parts)
    for b in (1, 255, 257, 700, 1000, 1576, 3200, 4096, 5000, 12345):
        parts = runtime2.plan_parts(b, 3)
        os.environ["CDX_UNET2_PLAN"] = "bulk"
        try:
            assert cost(parts) <= cost(runtime2.plan_parts(b, 3)) + 1e-9          # never worse than the older rule by the cost table
        finally:
            del os.environ["CDX_UNET2_PLAN"]
        assert sum(c for _, c, _ in parts) == b and parts[0][0] == 0 and all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(len(parts) - 1))




def test_every_launch_part_fits_the_program_it_runs_on():
    """with a compact program present plan_for() cuts batches with T up to 3; a cut into T = 2 parts used to be launched on
    the DEFAULT program even when two of its LDS plans exceed 160 KiB (model_dim 32, dim_mult [1,2,4], H = 32: 90.7 KB each)."""
    net = JannerUNet1d(64, model_dim=32, dim_mult=[1, 2, 4])
    for b in (1, 256, 300, 512, 600, 768, 1024, 3200):
        comp, parts = runtime2.plan_for(net, 32, b)
        assert sum(c for _, c, _ in parts) == b
        assert all(comp.prog.lds_bytes(t) <= 160 * 1024 for _, _, t in parts), (b, parts)


def test_standalone_forward_of_a_v2_net_keeps_the_gemm_crossover():
    """the v2 kernel serves sampling loops only; a stand-alone forward (per-sample timesteps) of the same net follows the
    batch crossover between the program kernel and the implicit-GEMM executor."""
    net = JannerUNet1d(23, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5)
    assert not bigbatch.is_chiunet_gemm(net, 3200, 32)                       # loop: v2 program kernel
    assert bigbatch.is_chiunet_gemm(net, 3200, 32, forward=True)             # forward at the Diffuser batch: GEMM executor
    assert not bigbatch.is_chiunet_gemm(net, 256, 32, forward=True)


def test_sharded_sample_draws_the_solver_default_number_of_steps(monkeypatch):
    """sharded_sample(seed=...) without sample_steps follows the agent's own default (5 for the SDE solvers), not
    diffusion_steps (1000 -> 1001 global-batch draws)."""
    from cleandiffuser_amd import distributed
    from cleandiffuser_amd.diffusion import DiscreteDiffusionSDE
    from cleandiffuser_amd.diffusion.ddpm import DDPM
    seen = []
    monkeypatch.setattr(distributed, "global_noise", lambda shape, n, seed: seen.append(n) or [torch.zeros(shape) for _ in range(n)])
    net = JannerUNet1d(6, model_dim=16, emb_dim=16, dim_mult=[1, 2], kernel_size=5)
    distributed.sharded_sample(DiscreteDiffusionSDE(net, None, diffusion_steps=1000), torch.zeros(2, 8, 6), seed=1, solver="ddim")
    distributed.sharded_sample(DDPM(net, None, diffusion_steps=7), torch.zeros(2, 8, 6), seed=1)
    assert seen == [6, 8]


def test_weight_signature_sees_every_kind_of_change():
    """runtime._signature keys every packed-weight / program cache.  It no longer walks the module tree per call (round 4: that walk was
    1.6 of the 1.7 ms of host time of a steady-state sample() call), so every way weights can change must still move it: an in-place
    update (version counter), a `.data` swap (pointer), a replaced Parameter object, a replaced / added submodule, a buffer update, the
    explicit epoch -- and it must NOT move when nothing changed."""
    import torch
    from cleandiffuser_amd.engine import runtime as R
    from cleandiffuser_amd.nn_diffusion import JannerUNet1d
    from cleandiffuser_amd.utils import invalidate_weights
    net = JannerUNet1d(6, model_dim=16, emb_dim=16, dim_mult=[1, 2], kernel_size=5)
    s0 = R._signature(net)
    assert R._signature(net) == s0 and len(s0) == 1 + len(list(net.parameters())) + len(list(net.buffers()))
    seen = {s0}

    def changed(what):
        s = R._signature(net)
        assert s not in seen, what
        assert R._signature(net) == s, what + " (not stable)"
        seen.add(s)
    with torch.no_grad():
        next(net.parameters()).add_(1.0)
    changed("in-place update")
    p = list(net.parameters())[3]
    p.data = p.data.clone()
    changed(".data swap")
    net.final_conv[3].weight = torch.nn.Parameter(net.final_conv[3].weight.detach().clone())
    changed("replaced Parameter")
    net.final_conv[3] = torch.nn.Conv1d(16, 6, 1)
    changed("replaced submodule")
    net.extra = torch.nn.Linear(2, 2)
    changed("added submodule")
    net.register_buffer("stat", torch.zeros(3))
    changed("added buffer")
    net.stat.add_(1.0)
    changed("buffer update")
    invalidate_weights(net)
    changed("explicit epoch")
