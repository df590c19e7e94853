"""Plain-torch stand-ins for the kernel wrappers of cleandiffuser_amd/engine/blocks.py -- TEST HELPER for the CPU tier.

The training path (engine/train.py) is ~900 lines of HOST logic around the HIP kernels: which node runs which launch with which layout,
how gradients are routed (autograd, or straight into ``.grad``), when the weight layouts are refreshed.  None of it can execute without a
GPU -- unless the dozen kernel wrappers it calls are replaced by torch expressions of the same contracts, which is what ``emulated()``
does (monkeypatch-style, restored on exit).  The KERNELS are checked on the MI355X (tests/test_gpu_parity.py); this checks everything
around them, against torch.autograd of the modules' own PyTorch forward."""
import contextlib

import torch
import torch.nn.functional as F

from cleandiffuser_amd.engine import blocks, train

ACTS = {"mish": F.mish, "gelu": F.gelu, "gelu_tanh": lambda z: F.gelu(z, approximate="tanh"), "none": lambda z: z, "silu": F.silu,
        "leaky": F.leaky_relu, "relu": F.relu, "tanh": torch.tanh}


def _vjp(fn, inputs, dy):
    leaves = [t.detach().clone().requires_grad_(True) for t in inputs]
    with torch.enable_grad():
        fn(*leaves).backward(dy)
    return [t.grad for t in leaves]


def linear(a, w, bias=None, out=None, partial=None, **kw):
    assert not kw.get("act") or kw["act"] == "none"
    y = a @ w.t()
    return y + bias if bias is not None else y


def conv1d(x, w_packed, bias, batch, l_in, stride=1, pad=0, out=None, residual=None, act="none", l_out=None, partial=None):
    n, taps, cin = w_packed.shape
    if l_out is None:
        l_out = (l_in + 2 * pad - taps) // stride + 1
    right = (l_out - 1) * stride + taps - l_in - pad
    xc = x.reshape(batch, l_in, cin).permute(0, 2, 1)
    xc = F.pad(xc, (pad, max(right, 0)))
    y = F.conv1d(xc, w_packed.permute(0, 2, 1), bias, stride=stride)[:, :, :l_out].permute(0, 2, 1).reshape(batch * l_out, n)
    assert act == "none"
    if residual is not None:                              # epilogue operand of cdx_gemm_f32: one fp32 add per element
        y = y + residual
    if out is not None:
        out.copy_(y)
        return out
    return y


def _gn(x, gamma, beta, batch, length, groups, act, eps):
    c = x.shape[1]
    y = F.group_norm(x.reshape(batch, length, c).permute(0, 2, 1), groups, gamma, beta, eps).permute(0, 2, 1).reshape(batch * length, c)
    return ACTS[act](y)


def groupnorm(x, gamma, beta, batch, length, groups, act="none", eps=1e-5, fb=None, film_mode=0, residual=None, **kw):
    """cdx_groupnorm_f32: y = act(gn(x)) [+ fb[b] (film_mode 2: a per-sample additive vector)] [+ residual]."""
    assert all(v is None or v == 0 or v is False for v in kw.values()), kw
    y = _gn(x, gamma, beta, batch, length, groups, act, eps)
    if fb is not None:
        assert film_mode == 2 and fb.shape == (batch, x.shape[1])
        y = (y.view(batch, length, -1) + fb[:, None, :]).reshape(batch * length, -1)
    if residual is not None:
        y = y + residual
    return y


def groupnorm_backward(dy, x, gamma, beta, batch, length, groups, act="mish", eps=1e-5, out=None, param_grads=False, grads_out=None,
                       possum_out=None):
    dx, dg, db = _vjp(lambda a, g, b: _gn(a, g, b, batch, length, groups, act, eps), (x, gamma, beta), dy)
    if possum_out is not None:                            # cdx_gn_args.dy_possum: dy summed over each sample's positions
        possum_out.copy_(dy.view(batch, length, -1).sum(1))
    if not param_grads:
        return dx
    if grads_out is not None:
        grads_out[0].add_(dg)
        grads_out[1].add_(db)
        return dx, None, None
    return dx, dg, db


def colsum(x, out=None):
    if out is None:
        return x.sum(0)
    out.add_(x.sum(0))
    return out


def conv_wgrad(p, q, batch, l_p, l_q, taps, stride=1, pad=0, k_split=0, bias_grad=False, dw_out=None, db_out=None):
    ca, cb = p.shape[1], q.shape[1]
    p3, q3 = p.reshape(batch, l_p, ca), q.reshape(batch, l_q, cb)
    dw = torch.zeros(ca, cb, taps)
    m = torch.arange(l_p)
    for t in range(taps):
        idx = m * stride + t - pad
        ok = (idx >= 0) & (idx < l_q)
        dw[:, :, t] = torch.einsum("nma,nmb->ab", p3[:, ok], q3[:, idx[ok]])
    db = p.sum(0)
    if dw_out is not None:
        dw_out.add_(dw.view(dw_out.shape))
        if bias_grad:
            db_out.add_(db)
        return (dw_out, db_out) if bias_grad else dw_out
    return (dw, db) if bias_grad else dw


def conv_wgrad_batch(jobs):
    """blocks.conv_wgrad_batch: the queued products of a backward pass, added into their ``.grad`` homes."""
    COUNTS["wgrad_batch"] = COUNTS.get("wgrad_batch", 0) + 1
    for p, q, batch, l_p, l_q, taps, stride, pad, dw, db in jobs:
        conv_wgrad(p, q, batch, l_p, l_q, taps, stride, pad, bias_grad=db is not None, dw_out=dw, db_out=db)
    return -(-len(jobs) // 32)


def activation(z, act, out=None):
    return ACTS[act](z)


def activation_backward(pre, g, act, param=1.0, out=None):
    return _vjp(ACTS[act], (pre,), g)[0]


def _ln(x, gamma, beta, scale, shift, rows_per_mod, eps):
    y = F.layer_norm(x, (x.shape[1],), gamma, beta, eps)
    if scale is not None:
        b = x.shape[0] // rows_per_mod
        y = (y.view(b, rows_per_mod, -1) * (1 + scale[:, None]) + shift[:, None]).reshape(x.shape)
    return y


def layernorm(x, out=None, gamma=None, beta=None, scale=None, shift=None, rows_per_mod=1, eps=1e-5):
    return _ln(x, gamma, beta, scale, shift, rows_per_mod, eps)


def layernorm_backward(dy, x, gamma=None, scale=None, rows_per_mod=1, eps=1e-5, want_dyxhat=False):
    shift = None if scale is None else torch.zeros_like(scale)
    dx = _vjp(lambda a: _ln(a, gamma, None if gamma is None else torch.zeros_like(gamma), scale, shift, rows_per_mod, eps), (x,), dy)[0]
    return (dx, dy * F.layer_norm(x, (x.shape[1],), None, None, eps)) if want_dyxhat else dx


def _core(q, k, v, batch, n_heads, mask, keep):
    dm = q.shape[1]
    dh = dm // n_heads
    qh, kh, vh = (z.reshape(batch, -1, n_heads, dh).transpose(1, 2) for z in (q, k, v))
    sc = qh @ kh.transpose(-1, -2) / dh ** 0.5
    p = torch.softmax(sc if mask is None else sc + mask, -1)
    return ((p if keep is None else p * keep) @ vh).transpose(1, 2).reshape(-1, dm)


def mha_train(q, k, v, batch, n_heads, mask=None, keep=None, dout=None, grads=None):
    assert q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1
    if dout is None:
        return _core(q, k, v, batch, n_heads, mask, keep)
    got = _vjp(lambda a, b, c: _core(a, b, c, batch, n_heads, mask, keep), (q, k, v), dout)
    if grads is None:
        return tuple(got)
    for dst, g in zip(grads, got):
        dst.copy_(g)
    return grads


def attention(qkv, batch, tokens, n_heads, out=None, mask=None):
    dm = qkv.shape[1] // 3
    return _core(qkv[:, :dm], qkv[:, dm:2 * dm], qkv[:, 2 * dm:], batch, n_heads, mask, None)


def attention_backward(qkv, dout, batch, tokens, n_heads):
    return _vjp(lambda z: attention(z, batch, tokens, n_heads), (qkv,), dout)[0]


def relayout_table(jobs, device):
    return (list(jobs), None, len(jobs))


def relayout(table):
    COUNTS["relayout"] += 1
    for src, off, dst, n, st in table[0]:
        store = torch.as_strided(src, (src.untyped_storage().nbytes() // 4,), (1,), 0)
        i0, i1, i2 = torch.meshgrid(torch.arange(n[0]), torch.arange(n[1]), torch.arange(n[2]), indexing="ij")
        dst.view(-1).copy_(store[src.storage_offset() + off + i0 * st[0] + i1 * st[1] + i2 * st[2]].reshape(-1))


COUNTS = {"relayout": 0, "aten_pack": 0}
_NAMES = ("linear", "conv1d", "groupnorm", "groupnorm_backward", "colsum", "conv_wgrad", "conv_wgrad_batch", "activation", "activation_backward", "layernorm",
          "layernorm_backward", "mha_train", "attention", "attention_backward", "relayout_table", "relayout")


@contextlib.contextmanager
def emulated():
    saved = {n: getattr(blocks, n) for n in _NAMES}
    scratch, aten_pack, capturing = train._splitk_scratch, train._aten_pack, torch.cuda.is_current_stream_capturing
    for n in _NAMES:
        setattr(blocks, n, globals()[n])
    train._splitk_scratch = lambda *a: None

    def counted(kind, w):
        t = aten_pack(kind, w)
        COUNTS["aten_pack"] += int(t.untyped_storage().data_ptr() != w.untyped_storage().data_ptr())      # (views of the weight are free)
        return t
    train._aten_pack = counted
    torch.cuda.is_current_stream_capturing = lambda: False
    COUNTS.update(relayout=0, aten_pack=0)
    try:
        yield COUNTS
    finally:
        for n, f in saved.items():
            setattr(blocks, n, f)
        train._splitk_scratch, train._aten_pack, torch.cuda.is_current_stream_capturing = scratch, aten_pack, capturing
