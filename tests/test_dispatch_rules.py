"""Host-side routing rules between the two native U-Net executors and the PyTorch executor (engine/dispatch.py, bigbatch.py,
runtime.py) -- pure Python decisions, checked on the CPU."""
import torch

from cleandiffuser_amd.engine import bigbatch, plan as P, program, runtime, runtime2
from cleandiffuser_amd.nn_diffusion import ChiUNet1d, DiT1d, JannerUNet1d


def test_unet_executor_choice(monkeypatch):
    janner = JannerUNet1d(6, model_dim=16, emb_dim=16, dim_mult=[1, 2], kernel_size=5)
    chi = ChiUNet1d(2, 5, 1, model_dim=32, emb_dim=32, dim_mult=[1, 2], obs_as_global_cond=True)
    chi_local = ChiUNet1d(2, 5, 1, model_dim=32, emb_dim=32, dim_mult=[1, 2], obs_as_global_cond=False)
    assert not bigbatch.is_chiunet_gemm(janner, 256) and bigbatch.is_chiunet_gemm(janner, bigbatch.JANNER_GEMM_MIN_BATCH)
    assert not bigbatch.is_chiunet_gemm(chi, 8) and bigbatch.is_chiunet_gemm(chi, bigbatch.UNET_GEMM_MIN_BATCH)
    # local conditioning: the implicit-GEMM executor is its only native path, at every batch size
    assert bigbatch.is_chiunet_gemm(chi_local, 1) and bigbatch.is_chiunet_gemm(chi_local, 10 ** 6)
    assert not bigbatch.is_chiunet_gemm(DiT1d(4, 8, d_model=16, n_heads=2, depth=1), 10 ** 6)
    seen = []

    def fake_supported(module, horizon, edm=False):
        seen.append((horizon, edm))
        return "LDS plan needs 200000 B" if (horizon >= 64 or edm) else None
    monkeypatch.setattr(runtime, "supported_backbone", fake_supported)
    # the second-generation program kernel has its own (smaller) LDS plan: pretend it shares the fake limit of the first
    monkeypatch.setattr(runtime2, "supported", lambda module, horizon: "LDS plan needs 200000 B" if horizon >= 64 else None)
    monkeypatch.setattr(runtime2, "compact_only", lambda module, horizon: False)
    assert not bigbatch.is_chiunet_gemm(janner, 3, 32)                  # fits the program kernel: small batches stay there
    assert bigbatch.is_chiunet_gemm(janner, 3, 64)                      # does not fit: GEMM executor at any batch
    assert bigbatch.is_chiunet_gemm(chi, 3, 16, True)                   # fits only without the EDM buffers, plan has EDM steps
    assert seen == [(64, False), (16, True)]                            # (v2 answered for the H = 32 request)
    assert not bigbatch.is_chiunet_gemm(janner, 5000, 32)               # v2 program kernel keeps every batch size it can run
    assert bigbatch.is_chiunet_gemm(janner, 5000, 32, True) and bigbatch.is_chiunet_gemm(janner, 5000, 64)   # EDM plans / too big: GEMM
    assert len(seen) == 2                                               # large batch: no need to ask the v1 compiler


def test_edm_state_buffers_are_optional_in_the_lds_plan():
    net = JannerUNet1d(23, model_dim=32, emb_dim=32, dim_mult=[1, 2, 2, 2], kernel_size=5)
    lean, full = program.compile_janner(net, 32, edm=False), program.compile_janner(net, 32, edm=True)
    assert full.lds_floats - lean.lds_floats == 2 * ((32 * 23 + 3) // 4 * 4)
    assert (lean.prev_off, lean.x_off, lean.pred_off) == (full.prev_off, full.x_off, full.pred_off)
    assert len(lean.ops) == len(full.ops) and lean.macs_per_forward == full.macs_per_forward
    euler = P.SamplePlan(solver="x", steps=[P.Step(P.KIND_EDM_EULER, P.V_EPS, 1, 0.1, 1.0, 1.0, (1.0, 1.0, 1.0, 0.1, 0.0))])
    ddim = P.SamplePlan(solver="ddim", steps=[P.Step(P.KIND_DDIM, P.V_EPS, 1, 3, 0.9, 0.4, (1.0, 0.4, 0.9, 0.0, 0.0))])
    assert runtime.plan_is_edm(euler) and not runtime.plan_is_edm(ddim)
    # the shipped Diffuser kitchen net: 159.7 KB without, 177 KB with -- the difference between fused and not
    kitchen = JannerUNet1d(69, model_dim=64, emb_dim=64, dim_mult=[1, 2, 2, 2], kernel_size=5)
    assert program.compile_janner(kitchen, 32, edm=False).lds_floats * 4 <= 160 * 1024
    try:
        program.compile_janner(kitchen, 32, edm=True)
        raise AssertionError("expected the EDM variant not to fit")
    except ValueError as e:
        assert "LDS plan" in str(e)


def test_launch_plan_cuts_large_batches():
    """B = 3200 = 4 rounds of 768 trajectories three per workgroup + 128 one per workgroup; B = 512 one round of two; B = 256 one of one."""
    from cleandiffuser_amd.engine import runtime2
    assert runtime2.plan_parts(3200, 3) == [(0, 3072, 3), (3072, 128, 1)]
    assert runtime2.plan_parts(512, 3) == [(0, 512, 2)]
    assert runtime2.plan_parts(256, 3) == [(0, 256, 1)]
    assert runtime2.plan_parts(640, 3) == [(0, 640, 3)]
    assert runtime2.plan_parts(3200, 2) == [(0, 3072, 2), (3072, 128, 1)]
    for b in (1, 255, 257, 700, 1000, 5000):
        parts = runtime2.plan_parts(b, 3)
        assert sum(c for _, c, _ in parts) == b and parts[0][0] == 0 and all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(len(parts) - 1))


