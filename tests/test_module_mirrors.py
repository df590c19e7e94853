"""Module mirrors that no sampling fixture exercises, against outputs of the REAL reference (tests/golden/modules.npz, produced by
oracle/gen_module_golden.py): extra backbones, classifier networks, classifier wrappers (logp, input gradients, QGPO loss)."""
import numpy as np
import pytest
import torch

from conftest import golden_path
from oracle import gen_module_golden as G


@pytest.mark.parametrize("name", list(G.specs()))
def test_module_forward_matches_reference(name):
    gold = np.load(golden_path("modules"))
    net, args = G.build("cleandiffuser_amd", name)
    with torch.no_grad():
        y = net(*args).numpy()
    np.testing.assert_allclose(y, gold[name], rtol=2e-6, atol=2e-6)


def test_classifier_wrappers_match_reference():
    gold = np.load(golden_path("modules"))
    out = G.wrapper_outputs("cleandiffuser_amd")
    for k, v in out.items():
        np.testing.assert_allclose(v, gold[k], rtol=2e-5, atol=2e-6, err_msg=k)


def test_post_sampling_heads_match_reference():
    """Critics, inverse dynamics, transformer toolkit on CPU against the reference's outputs."""
    gold = np.load(golden_path("modules"))
    out = G.head_outputs("cleandiffuser_amd")
    # (the reference's Decision-Veteran critic / transformer toolkit are not mirrored: their fixture entries stay reference-only)
    ref_only = {"head/DVHorizonCritic/post", "head/DVHorizonCritic/pre", "head/Transformer/y", "head/Transformer/map1"}
    assert {f"head/{k}" for k in out} == {k for k in gold.files if k.startswith("head/")} - ref_only
    for k, v in out.items():
        np.testing.assert_allclose(v, gold[f"head/{k}"], rtol=2e-6, atol=2e-6, err_msg=k)


def test_head_chain_compiler_covers_every_head():
    """engine/heads.py must understand every Sequential the heads are made of (otherwise the GPU path silently runs eager)."""
    import torch.nn as nn
    from cleandiffuser_amd.engine import heads
    from cleandiffuser_amd.invdynamic.mlp import EnsembleMlpInvDynamic, FancyMlpInvDynamic, MlpInvDynamic
    from cleandiffuser_amd.utils import DQLCritic, TwinQ, V
    seqs = [DQLCritic(5, 2, 16).q1_model, TwinQ(5, 2, 16).Q2, V(5, 16).V, MlpInvDynamic(5, 2, 16).mlp.mlp,
            FancyMlpInvDynamic(5, 2, 16, add_norm=True, add_dropout=True).model.eval(),
            EnsembleMlpInvDynamic(5, 2, 16, n_models=2, mlp_type="fancy").mlp[0].eval()]
    for s in seqs:
        ops = heads.compile_chain(s)
        assert ops is not None
        assert sum(o[0] == "linear" for o in ops) == sum(isinstance(m, nn.Linear) for m in s.modules())
        assert not any(o[0] == "act" for o in ops), "every activation should fuse into the launch before it"
    ops = heads.compile_chain(DQLCritic(5, 2, 16).q1_model)
    assert [o[0] for o in ops] == ["linear", "norm"] * 3 + ["linear"] and [o[2] for o in ops[1::2]] == ["tanh", "mish", "mish"]
    fancy = FancyMlpInvDynamic(5, 2, 16, add_dropout=True).model.train()
    assert heads.compile_chain(fancy) is None            # active dropout: stock modules
    assert heads.compile_chain(nn.Sequential(nn.Linear(3, 3), nn.Softplus())) is None


def test_edm_variants_match_reference():
    """VPODE / VEODE / EDMDDIM: tables, preconditioning, loss equal to the reference; sample() fails the way the reference's does."""
    import torch
    gold = np.load(golden_path("modules"))
    out = G.edm_variant_outputs("cleandiffuser_amd")
    for k, v in out.items():
        np.testing.assert_allclose(v, gold[f"edmvar/{k}"], rtol=1e-6, atol=1e-7, err_msg=k)
    from cleandiffuser_amd.diffusion.veode import VEODE
    from cleandiffuser_amd.nn_diffusion import DQLMlp
    agent = VEODE(DQLMlp(5, 3, emb_dim=16), None, diffusion_steps=50)
    agent.eval()
    with pytest.raises(IndexError):
        agent.sample(torch.zeros(4, 3), solver="euler", n_samples=4, sample_steps=6)


def test_head_chain_semantics_with_torch_stand_ins(monkeypatch):
    """engine/heads.py end to end on the CPU: the three block launchers are replaced by PyTorch stand-ins with the same contract
    (linear: act(x W^T + b); groupnorm with L = 1, G = 1: per-row LayerNorm, affine, activation), so the compiled chain, the
    activation-name table, the leading-dimension handling and the cache invalidation are checked without a GPU."""
    import torch.nn as nn
    import torch.nn.functional as F
    from cleandiffuser_amd.engine import blocks, heads
    from cleandiffuser_amd.invdynamic.mlp import EnsembleMlpInvDynamic, FancyMlpInvDynamic, MlpInvDynamic
    from cleandiffuser_amd.nn_condition import MLPCondition, PearceObsCondition
    from cleandiffuser_amd.utils import DQLCritic, TwinQ, V, load_synth
    acts = {"none": lambda v: v, "relu": F.relu, "tanh": torch.tanh, "mish": F.mish, "silu": F.silu, "gelu": F.gelu,
            "gelu_tanh": lambda v: F.gelu(v, approximate="tanh"), "leaky": lambda v: F.leaky_relu(v, 0.01)}
    monkeypatch.setattr(blocks, "linear", lambda h, w, b=None, act="none", **k: acts[act](F.linear(h, w, b)))
    monkeypatch.setattr(blocks, "groupnorm", lambda h, g, b, batch, length, groups, act="none", eps=1e-5, **k:
                        acts[act](F.layer_norm(h, (h.shape[1],), g, b, eps)) if (length, groups, batch) == (1, 1, h.shape[0]) else None)
    monkeypatch.setattr(blocks, "activation", lambda h, act, **k: acts[act](h))
    monkeypatch.setattr(heads, "native_ok", lambda x, params: True)
    torch.manual_seed(0)
    with torch.no_grad():
        for seq, x in [(load_synth(DQLCritic(5, 2, 16)).q2_model.eval(), torch.randn(9, 7)),
                       (load_synth(TwinQ(5, 2, 24)).Q1.eval(), torch.randn(4, 3, 7)),                   # leading dims kept
                       (load_synth(V(5, 16)).V.eval(), torch.randn(1, 5)),
                       (MlpInvDynamic(5, 2, 16).mlp.mlp.eval(), torch.randn(6, 10)),
                       (FancyMlpInvDynamic(5, 2, 16, add_norm=True, add_dropout=True).model.eval(), torch.randn(6, 10)),
                       (EnsembleMlpInvDynamic(5, 2, 16, n_models=2, mlp_type="fancy").mlp[1].eval(), torch.randn(6, 10)),
                       (load_synth(MLPCondition(1, 8, [8], nn.SiLU())).mlp.mlp.eval(), torch.rand(5, 1)),
                       (load_synth(PearceObsCondition(4, 8)).mlp.eval(), torch.randn(3, 2, 4)),
                       (nn.Sequential(nn.GELU(approximate="tanh"), nn.Linear(4, 3, bias=False), nn.LeakyReLU()).eval(), torch.randn(5, 4))]:
            want, got = seq(x), heads.try_sequential(seq, x)
            assert got is not None and got.shape == want.shape
            np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
        seq = FancyMlpInvDynamic(5, 2, 16, add_dropout=True).model.eval()
        x = torch.randn(6, 10)
        assert heads.try_sequential(seq, x) is not None
        seq.train()                                                   # dropout becomes active: recompiled, refused
        assert heads.try_sequential(seq, x) is None
        seq.eval()
        seq[4] = nn.Linear(16, 16)                                    # swapped sub-module: recompiled against the new one
        np.testing.assert_allclose(heads.try_sequential(seq, x).numpy(), seq(x).numpy(), rtol=1e-6, atol=1e-6)
        assert heads.try_sequential(seq, torch.zeros(0, 10)) is None   # empty batch: stock modules


def test_pearce_transformer_folding_equals_the_module():
    """cdx_pearcetf_run consumes folded weights (engine/bigbatch.py:fold_pearcetf: in_proj o input_to_qkv1, attn1_to_fcn o out_proj,
    eval-mode BatchNorm1d + 1/1.414 residual scaling, position codes in the token biases).  The executor's data flow re-stated with
    torch ops on those folded tensors must equal the module's own forward (reference pearcetransformer.py:91-151)."""
    import torch
    import torch.nn.functional as F
    from cleandiffuser_amd.engine.bigbatch import fold_pearcetf
    from cleandiffuser_amd.nn_diffusion import PearceTransformer
    from cleandiffuser_amd.utils import load_synth
    net = load_synth(PearceTransformer(6, To=2, emb_dim=32, trans_emb_dim=16, nhead=4), 9).eval()
    for m in net.modules():                                  # non-trivial running statistics
        if isinstance(m, torch.nn.BatchNorm1d):
            m.running_mean.uniform_(-0.5, 0.5)
            m.running_var.uniform_(0.5, 2.0)
    fold = fold_pearcetf(net)
    g = torch.Generator().manual_seed(0)
    x, t, cond = torch.randn(7, 6, generator=g), torch.arange(7) * 3, torch.randn(7, 2, 32, generator=g)
    with torch.no_grad():
        want = net._forward_torch(x, t, cond) if hasattr(net, "_forward_torch") else net(x, t, cond)
        te, td, H = fold["te"], fold["td"], fold["heads"]
        xi = F.linear(net.act_emb(x), net.act_to_input.weight, fold["a2i_b"])
        ti = F.linear(net.map_noise(t), net.t_to_input.weight, fold["t2i_b"])
        ci = net.cond_to_input(cond) + fold["cpos"][None]
        f = torch.cat([xi[:, None], ti[:, None], ci], 1)                       # (b, S, te), batch-major rows as the executor keeps them
        b, S = f.shape[:2]
        for k in fold["blocks"]:
            qkv = F.linear(f, k["qkv_w"], k["qkv_b"]).reshape(b, S, 3, H, te)
            q, kk, v = (qkv[:, :, j].permute(0, 2, 1, 3) for j in range(3))     # (b, H, S, te)
            att = torch.softmax(q @ kk.transpose(-1, -2) / te ** 0.5, -1) @ v
            att = att.permute(0, 2, 1, 3).reshape(b, S, td)
            a1 = F.linear(att, k["o_w"], k["o_b"]) + f * k["r1"]
            f = F.linear(F.gelu(F.linear(a1, k["fc1_w"], k["fc1_b"])), k["fc2_w"], k["fc2_b"]) + a1 * k["r2"]
        got = net.final(f.reshape(b, S * te))
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5)
    net.train()
    assert fold_pearcetf(net) is None                         # batch statistics: not this executor's business
