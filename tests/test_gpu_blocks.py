"""GPU unit tests of the big-batch building blocks (cdx_gemm_f32 / cdx_layernorm_f32 / cdx_attention_f32) against the
plain PyTorch fp32 ops they replace -- evaluated on the CPU in fp64 as the reference, so the bar is fp32 round-off."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _ref(t):
    return t.detach().cpu().double()


@pytest.mark.parametrize("m,n,k", [(1, 1, 1), (7, 29, 320), (300, 320, 29), (1024, 960, 320), (130, 129, 17),
                                   (4096, 1280, 320), (513, 320, 1280)])
def test_gemm_matches_linear(m, n, k):
    from cleandiffuser_amd.engine import blocks
    g = torch.Generator().manual_seed(m * 7 + n)
    a, w, b = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g)
    out = blocks.linear(a.to(DEV), w.to(DEV), b.to(DEV))
    ref = F.linear(_ref(a), _ref(w), _ref(b))
    torch.testing.assert_close(out.cpu().double(), ref, rtol=2e-5, atol=2e-5)


def test_gemm_asymmetric_identity_catches_transposes():
    from cleandiffuser_amd.engine import blocks
    a = torch.eye(160)
    w = torch.arange(96 * 160, dtype=torch.float32).reshape(96, 160) * 1e-3
    out = blocks.linear(a.to(DEV), w.to(DEV))
    torch.testing.assert_close(out.cpu(), w.t().contiguous(), rtol=0, atol=1e-6)


@pytest.mark.parametrize("act", ["none", "mish", "gelu", "gelu_tanh", "silu", "leaky", "relu", "tanh"])
def test_gemm_fused_epilogue(act):
    from cleandiffuser_amd.engine import blocks
    g = torch.Generator().manual_seed(3)
    B, T, K, N = 5, 16, 64, 96
    a, w, b = torch.randn(B * T, K, generator=g), torch.randn(N, K, generator=g) / 8, torch.randn(N, generator=g)
    gate, res, tab = torch.randn(B, N, generator=g), torch.randn(B * T, N, generator=g), torch.randn(T, N, generator=g)
    out = blocks.linear(a.to(DEV), w.to(DEV), b.to(DEV), act=act, gate=gate.to(DEV), rows_per_gate=T,
                        residual=res.to(DEV), table=tab.to(DEV))
    y = F.linear(_ref(a), _ref(w), _ref(b))
    fn = {"none": lambda v: v, "mish": F.mish, "gelu": F.gelu, "gelu_tanh": lambda v: F.gelu(v, approximate="tanh"),
          "silu": F.silu, "leaky": lambda v: F.leaky_relu(v, 0.01), "relu": F.relu, "tanh": torch.tanh}[act]
    ref = fn(y) * _ref(gate).repeat_interleave(T, 0) + _ref(res) + _ref(tab).repeat(B, 1)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=2e-5, atol=2e-5)


def test_gemm_strided_views_and_empty():
    from cleandiffuser_amd.engine import blocks
    g = torch.Generator().manual_seed(4)
    big = torch.randn(40, 96, generator=g).to(DEV)
    a = big[:, 32:64]                                   # row stride 96, 32 columns
    w = torch.randn(24, 32, generator=g).to(DEV)
    out = blocks.linear(a, w)
    torch.testing.assert_close(out.cpu().double(), F.linear(_ref(a), _ref(w)), rtol=2e-5, atol=2e-5)
    assert blocks.linear(torch.zeros(0, 32, device=DEV), w).shape == (0, 24)


@pytest.mark.parametrize("c", [64, 320, 1024, 100])
def test_layernorm_modulate(c):
    from cleandiffuser_amd.engine import blocks
    g = torch.Generator().manual_seed(c)
    B, T = 3, 8
    x = torch.randn(B * T, c, generator=g) * 3 + 1
    sc, sh = torch.randn(B, c, generator=g), torch.randn(B, c, generator=g)
    y = blocks.layernorm(x.to(DEV), scale=sc.to(DEV), shift=sh.to(DEV), rows_per_mod=T, eps=1e-6)
    ref = F.layer_norm(_ref(x), (c,), eps=1e-6) * (1 + _ref(sc).repeat_interleave(T, 0)) + _ref(sh).repeat_interleave(T, 0)
    torch.testing.assert_close(y.cpu().double(), ref, rtol=2e-5, atol=2e-5)
    ga, be = torch.randn(c, generator=g), torch.randn(c, generator=g)
    y2 = blocks.layernorm(x.to(DEV), gamma=ga.to(DEV), beta=be.to(DEV), eps=1e-5)
    torch.testing.assert_close(y2.cpu().double(), F.layer_norm(_ref(x), (c,), _ref(ga), _ref(be), 1e-5), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("tokens,heads,dh", [(64, 10, 32), (16, 4, 16), (5, 2, 64), (1, 1, 8), (64, 3, 64), (33, 5, 20), (7, 3, 6),
                                             (65, 2, 32), (96, 4, 32), (200, 3, 64), (128, 2, 6)])     # > 64: streamed-key kernel
def test_attention_matches_mha_core(tokens, heads, dh):
    from cleandiffuser_amd.engine import blocks
    g = torch.Generator().manual_seed(tokens)
    B, dm = 3, heads * dh
    torch.full((1 << 22,), float("nan"), device=DEV)     # whatever stale memory the kernel might touch is poisoned first
    qkv = torch.randn(B * tokens, 3 * dm, generator=g) * 2
    out = blocks.attention(qkv.to(DEV), B, tokens, heads)
    q, k, v = (_ref(qkv).reshape(B, tokens, 3, heads, dh)[:, :, i].transpose(1, 2) for i in range(3))
    ref = F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B * tokens, dm)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("tokens,heads,dh", [(16, 4, 64), (64, 2, 32), (12, 4, 16), (9, 3, 6), (100, 2, 32)])
def test_attention_with_additive_mask(tokens, heads, dh):
    """Causal mask built the way nn.Transformer builds it (0 / -inf, clamped to the fp32 floor by the binding)."""
    from cleandiffuser_amd.engine import blocks
    g = torch.Generator().manual_seed(tokens + dh)
    B, dm = 3, heads * dh
    qkv = torch.randn(B * tokens, 3 * dm, generator=g) * 2
    allowed = torch.tril(torch.ones(tokens, tokens)) == 1
    mask = torch.zeros(tokens, tokens).masked_fill(~allowed, float("-inf"))
    out = blocks.attention(qkv.to(DEV), B, tokens, heads, mask=mask.clamp_min(torch.finfo(torch.float32).min).to(DEV))
    q, k, v = (_ref(qkv).reshape(B, tokens, 3, heads, dh)[:, :, i].transpose(1, 2) for i in range(3))
    ref = F.scaled_dot_product_attention(q, k, v, attn_mask=mask.double()).transpose(1, 2).reshape(B * tokens, dm)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("tokens,n_obs,heads,dh,per_sample", [(16, 2, 4, 64, False), (12, 3, 4, 16, True), (5, 0, 2, 8, False),
                                                              (16, 15, 2, 4, False), (7, 2, 3, 6, True), (3, 1, 1, 256, False),
                                                              (33, 4, 5, 128, True), (1, 0, 7, 32, False)])
def test_cross_attention_against_short_memory(tokens, n_obs, heads, dh, per_sample):
    from cleandiffuser_amd.engine import blocks
    g = torch.Generator().manual_seed(tokens * 3 + n_obs)
    B, dm, S = 4, heads * dh, 1 + n_obs
    q = torch.randn(B * tokens, dm, generator=g)
    kv_shared = torch.randn(B if per_sample else 3, 2 * dm, generator=g)
    kv_rows = torch.randn(max(B * n_obs, 1), 2 * dm, generator=g)
    t_idx, s_idx = torch.meshgrid(torch.arange(tokens), torch.arange(S), indexing="ij")
    mask = torch.zeros(tokens, S).masked_fill(~(t_idx >= (s_idx - 1)), float("-inf"))
    out = blocks.cross_attention(q.to(DEV), kv_shared.to(DEV), kv_rows.to(DEV) if n_obs else None, B, tokens, n_obs, heads,
                                 shared_row=2, shared_per_sample=per_sample,
                                 mask=mask.clamp_min(torch.finfo(torch.float32).min).to(DEV))
    shared = _ref(kv_shared)[torch.arange(B)] if per_sample else _ref(kv_shared)[2].expand(B, -1)
    mem = torch.cat([shared[:, None], _ref(kv_rows)[:B * n_obs].reshape(B, n_obs, 2 * dm)], 1)       # (B, S, 2dm)
    k = mem[..., :dm].reshape(B, S, heads, dh).transpose(1, 2)
    v = mem[..., dm:].reshape(B, S, heads, dh).transpose(1, 2)
    qq = _ref(q).reshape(B, tokens, heads, dh).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qq, k, v, attn_mask=mask.double()).transpose(1, 2).reshape(B * tokens, dm)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("cin,cout,k,stride,pad,L", [(32, 64, 5, 1, 2, 16), (256, 256, 3, 2, 1, 8), (2, 48, 5, 1, 2, 16),
                                                     (23, 32, 5, 1, 2, 32), (64, 23, 1, 1, 0, 4), (48, 40, 3, 1, 1, 1)])
def test_implicit_gemm_conv1d(cin, cout, k, stride, pad, L):
    from cleandiffuser_amd.engine import blocks
    g = torch.Generator().manual_seed(cin + cout)
    B = 5
    x = torch.randn(B, cin, L, generator=g)
    conv = torch.nn.Conv1d(cin, cout, k, stride, pad)
    rows = x.permute(0, 2, 1).reshape(B * L, cin).contiguous().to(DEV)
    out = blocks.conv1d(rows, blocks.pack_conv(conv.weight).to(DEV), conv.bias.detach().to(DEV), B, L, stride, pad)
    ref = conv.double()(x.double()).permute(0, 2, 1).reshape(-1, cout)
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("c,L", [(64, 4), (32, 16), (20, 3)])
def test_implicit_gemm_conv_transpose(c, L):
    from cleandiffuser_amd.engine import blocks
    g = torch.Generator().manual_seed(c)
    B = 3
    x = torch.randn(B, c, L, generator=g)
    up = torch.nn.ConvTranspose1d(c, c, 4, 2, 1)
    rows = x.permute(0, 2, 1).reshape(B * L, c).contiguous().to(DEV)
    packed = tuple(t.to(DEV) for t in blocks.pack_conv_transpose_k4s2p1(up.weight))
    out = blocks.conv_transpose1d_k4s2p1(rows, packed, up.bias.detach().to(DEV), B, L)
    ref = up.double()(x.double()).permute(0, 2, 1).reshape(-1, c)
    torch.testing.assert_close(out.cpu().double(), ref.detach(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("c,groups,L,film", [(256, 8, 16, 1), (1024, 8, 4, 1), (32, 8, 32, 2), (48, 4, 5, 0), (4096, 2, 16, 1)])
def test_groupnorm_mish_film_residual(c, groups, L, film):
    from cleandiffuser_amd.engine import blocks
    g = torch.Generator().manual_seed(c + L)
    B = 3
    x = torch.randn(B, c, L, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(c, generator=g), torch.randn(c, generator=g)
    width = {0: c, 1: 2 * c, 2: c}[film]
    fa, fb = torch.randn(4, width, generator=g), torch.randn(B, width, generator=g)
    res = torch.randn(B * L, c, generator=g)
    rows = x.permute(0, 2, 1).reshape(B * L, c).contiguous()
    out = blocks.groupnorm(rows.to(DEV), gamma.to(DEV), beta.to(DEV), B, L, groups, act="mish", fa=fa.to(DEV) if film else None,
                           fb=fb.to(DEV) if film else None, fa_row=2, film_mode=film, residual=res.to(DEV))
    y = F.mish(F.group_norm(_ref(x), groups, _ref(gamma), _ref(beta), 1e-5))                       # (B, C, L)
    f = _ref(fa)[2][None] + _ref(fb)
    if film == 1:
        y = f[:, :c, None] * y + f[:, c:, None]
    elif film == 2:
        y = y + f[:, :, None]
    ref = y.permute(0, 2, 1).reshape(B * L, c) + _ref(res)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=3e-5, atol=3e-5)


@pytest.mark.parametrize("m,n,k,slices", [(256, 256, 4096, 6), (300, 130, 2048, 4), (128, 64, 512, 8)])
def test_gemm_split_k_is_exact_and_deterministic(m, n, k, slices):
    """Few tiles + long K: the library splits K over `slices` partial buffers and reduces them in slice order."""
    from cleandiffuser_amd.engine import blocks
    g = torch.Generator().manual_seed(k)
    a, w, b = torch.randn(m, k, generator=g), torch.randn(n, k, generator=g) / k ** 0.5, torch.randn(n, generator=g)
    res = torch.randn(m, n, generator=g)
    part = torch.full((slices * m * n,), float("nan"), device=DEV)
    kw = dict(act="mish", residual=res.to(DEV), partial=part)
    o1 = blocks.linear(a.to(DEV), w.to(DEV), b.to(DEV), **kw)
    o2 = blocks.linear(a.to(DEV), w.to(DEV), b.to(DEV), **kw)
    assert torch.equal(o1, o2)
    ref = F.mish(F.linear(_ref(a), _ref(w), _ref(b))) + _ref(res)
    torch.testing.assert_close(o1.cpu().double(), ref, rtol=2e-5, atol=2e-5)
