"""Classifier guidance without autograd: ``d logp / d x`` of ``HalfJannerUNet1d`` by explicit forward + backward kernels.

What it replaces: ``BaseClassifier.gradients`` (reference classifier/base.py:74-79) -> ``torch.autograd.grad(logp.sum(), x)``
through ``HalfJannerUNet1d.forward`` (reference nn_classifier/half_jannerunet.py:102-125), called once per denoising step by
``classifier_guidance`` (diffusionsde.py:153-173) in every shipped Diffuser configuration (w_cg > 0).

How: channel-last activations ``(batch*L, C)``; every Conv1d is the implicit-GEMM conv of ``cdx_gemm_f32``; the backward of a
stride-1 conv is the same kernel with tap-flipped, transposed weights, the backward of the stride-2 downsample is two convs
writing even / odd rows; ``cdx_groupnorm_f32`` / ``cdx_groupnorm_bwd_f32`` do GroupNorm+Mish forward / backward; the forward
keeps the two pre-normalisation tensors of every residual block (the only state the backward needs).  Only the input gradient
is produced -- no weight gradients, no autograd graph.
"""
import ctypes
import weakref
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import blocks as B
from .runtime import _check, _f32c, _signature, _stream_ptr, load_library

_FP, _I = ctypes.c_void_p, ctypes.c_int32
ONE_CALL = True          # True: cdx_hjgrad_run sequences every launch in C; False: the same schedule issued from Python (debugging)


class CdxHjBlock(ctypes.Structure):
    _fields_ = [(n, _I) for n in ("cin", "cout", "k", "groups")] + \
               [(n, _FP) for n in ("w1", "b1", "w1_bwd", "g1", "be1", "w2", "b2", "w2_bwd", "g2", "be2", "emb_w", "emb_b",
                                   "wr", "br", "wr_bwd")]


class CdxHjDown(ctypes.Structure):
    _fields_ = [("c", _I)] + [(n, _FP) for n in ("w", "b", "bwd_even", "bwd_odd")]


class CdxHjgradWeights(ctypes.Structure):
    _fields_ = [(n, _I) for n in ("horizon", "in_dim", "model_dim", "emb_dim", "out_dim", "fc_hidden", "c_last", "l_last",
                                  "n_stages")] + \
               [("stage_kind", ctypes.POINTER(_I)), ("blocks", ctypes.POINTER(CdxHjBlock)), ("downs", ctypes.POINTER(CdxHjDown))] + \
               [(n, _FP) for n in ("map0_w", "map0_b", "map2_w", "map2_b", "fc1_wx", "fc1_wx_t", "fc1_we", "fc1_b", "fc2_w",
                                   "fc2_b", "fc2_w_t")]


_declared = False


def _lib():
    global _declared
    lib = load_library()
    if not _declared:
        lib.cdx_hjgrad_workspace_floats.argtypes = [ctypes.POINTER(CdxHjgradWeights), _I]
        lib.cdx_hjgrad_workspace_floats.restype = ctypes.c_longlong
        lib.cdx_hjgrad_run.argtypes = [ctypes.POINTER(CdxHjgradWeights), _FP, _FP, _I, _I, _FP, _FP, _FP, ctypes.c_longlong, _FP]
        lib.cdx_hjgrad_run.restype = ctypes.c_int
        _declared = True
    return lib


def _bwd_conv_s1(weight: torch.Tensor) -> torch.Tensor:
    """Conv1d(stride 1, pad k//2) weight (co, ci, k) -> packed backward-data kernel (ci, k, co): V[ci][u][co] = W[co][ci][k-1-u]."""
    return weight.detach().flip(2).permute(1, 2, 0).contiguous()


def _bwd_down_k3s2p1(weight: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Conv1d(k=3, stride 2, pad 1) backward-data as two stride-1 kernels over dY (one per parity of the input position):
       dX[2m]   = W[:, :, 1]^T dY[m]                        (1 tap, shift 0)
       dX[2m+1] = W[:, :, 2]^T dY[m] + W[:, :, 0]^T dY[m+1]  (2 taps, shifts 0, +1)"""
    w = weight.detach().permute(1, 2, 0)                       # (ci, 3, co)
    return w[:, [1]].contiguous(), w[:, [2, 0]].contiguous()


class _Block:
    def __init__(self, blk, dev):
        c1, gn1, c2, gn2 = blk.conv1[0], blk.conv1[1], blk.conv2[0], blk.conv2[1]
        f = lambda t: _f32c(t.detach(), dev)  # noqa: E731
        self.k, self.pad, self.groups, self.cout = c1.kernel_size[0], c1.padding[0], gn1.num_groups, c1.out_channels
        self.w1, self.b1, self.w1_b = f(B.pack_conv(c1.weight)), f(c1.bias), f(_bwd_conv_s1(c1.weight))
        self.w2, self.b2, self.w2_b = f(B.pack_conv(c2.weight)), f(c2.bias), f(_bwd_conv_s1(c2.weight))
        self.g1, self.be1, self.g2, self.be2 = f(gn1.weight), f(gn1.bias), f(gn2.weight), f(gn2.bias)
        self.emb_w, self.emb_b = f(blk.emb_mlp[1].weight), f(blk.emb_mlp[1].bias)
        self.has_res = isinstance(blk.residual_conv, nn.Conv1d)
        if self.has_res:
            self.wr, self.br = f(B.pack_conv(blk.residual_conv.weight)), f(blk.residual_conv.bias)
            self.wr_b = f(_bwd_conv_s1(blk.residual_conv.weight))


class _Down:
    def __init__(self, conv, dev):
        f = lambda t: _f32c(t.detach(), dev)  # noqa: E731
        self.w, self.b = f(B.pack_conv(conv.weight)), f(conv.bias)
        even, odd = _bwd_down_k3s2p1(conv.weight)
        self.even, self.odd = f(even), f(odd)


class HalfJannerGrad:
    """Bound (packed) weights of one HalfJannerUNet1d + the forward/backward schedule."""

    def __init__(self, net, dev):
        from ..utils import GroupNorm1d
        self.net, self.dev = net, dev
        self.ok = all(isinstance(m, GroupNorm1d) for blk in self._blocks(net) for m in (blk.conv1[1], blk.conv2[1]))
        if not self.ok:
            return
        self.stages = []                                        # [("block", _Block) | ("down", _Down)]
        for res1, res2, down in net.downs:
            self.stages += [("block", _Block(res1, dev)), ("block", _Block(res2, dev))]
            if not isinstance(down, nn.Identity):
                self.stages.append(("down", _Down(down.conv, dev)))
        for mid in (net.mid_block1, net.mid_block2):
            self.stages += [("block", _Block(mid[0], dev)), ("down", _Down(mid[1].conv, dev))]
        f = lambda t: _f32c(t.detach(), dev)  # noqa: E731
        self.map0_w, self.map0_b = f(net.map_emb[0].weight), f(net.map_emb[0].bias)
        self.map2_w, self.map2_b = f(net.map_emb[2].weight), f(net.map_emb[2].bias)
        fc1, fc2 = net.final_block[0], net.final_block[2]
        md = net.model_dim
        self.fc_in = fc1.in_features - md                       # C_last * L_last, reference order c * L + l
        self.c_last = self.stages[-2][1].cout
        self.l_last = self.fc_in // self.c_last
        wx = fc1.weight.detach()[:, :self.fc_in].reshape(-1, self.c_last, self.l_last).permute(0, 2, 1)   # -> [o][l][c]
        self.fc1_wx = f(wx.reshape(fc1.out_features, self.fc_in))
        self.fc1_wx_t = f(wx.reshape(fc1.out_features, self.fc_in).t())                                   # (fc_in, o)
        self.fc1_we, self.fc1_b = f(fc1.weight.detach()[:, self.fc_in:]), f(fc1.bias)
        self.fc2_w, self.fc2_b, self.fc2_w_t = f(fc2.weight), f(fc2.bias), f(fc2.weight.detach().t().contiguous())
        self._struct = self._bind(net, fc1, fc2)
        self._ws = None

    def _bind(self, net, fc1, fc2) -> CdxHjgradWeights:
        p = lambda t: t.data_ptr()  # noqa: E731
        blocks = [st for kind, st in self.stages if kind == "block"]
        downs = [st for kind, st in self.stages if kind == "down"]
        self._kinds = (_I * len(self.stages))(*[0 if kind == "block" else 1 for kind, _ in self.stages])
        self._blocks_c = (CdxHjBlock * len(blocks))(*[
            CdxHjBlock(cin=b.w1.shape[2], cout=b.cout, k=b.k, groups=b.groups, w1=p(b.w1), b1=p(b.b1), w1_bwd=p(b.w1_b), g1=p(b.g1),
                       be1=p(b.be1), w2=p(b.w2), b2=p(b.b2), w2_bwd=p(b.w2_b), g2=p(b.g2), be2=p(b.be2), emb_w=p(b.emb_w),
                       emb_b=p(b.emb_b), wr=p(b.wr) if b.has_res else None, br=p(b.br) if b.has_res else None,
                       wr_bwd=p(b.wr_b) if b.has_res else None) for b in blocks])
        self._downs_c = (CdxHjDown * max(len(downs), 1))(*[
            CdxHjDown(c=d.w.shape[0], w=p(d.w), b=p(d.b), bwd_even=p(d.even), bwd_odd=p(d.odd)) for d in downs])
        return CdxHjgradWeights(
            horizon=net.horizon, in_dim=net.in_dim, model_dim=net.model_dim, emb_dim=net.map_emb[0].in_features,
            out_dim=fc2.out_features, fc_hidden=fc1.out_features, c_last=self.c_last, l_last=self.l_last,
            n_stages=len(self.stages), stage_kind=self._kinds, blocks=self._blocks_c, downs=self._downs_c,
            map0_w=p(self.map0_w), map0_b=p(self.map0_b), map2_w=p(self.map2_w), map2_b=p(self.map2_b), fc1_wx=p(self.fc1_wx),
            fc1_wx_t=p(self.fc1_wx_t), fc1_we=p(self.fc1_we), fc1_b=p(self.fc1_b), fc2_w=p(self.fc2_w), fc2_b=p(self.fc2_b),
            fc2_w_t=p(self.fc2_w_t))

    def _one_call(self, x, emb0):
        lib = _lib()
        b = x.shape[0]
        need = lib.cdx_hjgrad_workspace_floats(ctypes.byref(self._struct), b)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(int(need), dtype=torch.float32, device=self.dev)
        logp = torch.empty((b, self._struct.out_dim), dtype=torch.float32, device=self.dev)
        grad = torch.empty_like(x)
        _check(lib.cdx_hjgrad_run(ctypes.byref(self._struct), x.data_ptr(), emb0.data_ptr(), emb0.shape[1], b, logp.data_ptr(), grad.data_ptr(),
                                  self._ws.data_ptr(), self._ws.numel(), _stream_ptr(self.dev)), "cdx_hjgrad_run")
        return logp, grad

    @staticmethod
    def _blocks(net):
        for res1, res2, _ in net.downs:
            yield res1
            yield res2
        yield net.mid_block1[0]
        yield net.mid_block2[0]

    def __call__(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor]):
        """x (b, H, D), noise (b,), condition (b, emb_dim)|None -> (logp (b, out_dim), d logp.sum() / d x (b, H, D))."""
        net, dev = self.net, self.dev
        b, H, D = x.shape
        with torch.no_grad():
            emb0 = _f32c(net.map_noise(noise), dev)
            if condition is not None:
                emb0 = emb0 + condition
            if ONE_CALL and H == net.horizon:
                return self._one_call(_f32c(x, dev), _f32c(emb0, dev))
            emb = B.linear(B.linear(emb0, self.map0_w, self.map0_b, act="mish"), self.map2_w, self.map2_b)     # (b, md)
            memb = B.activation(emb, "mish")
            cur = _f32c(x, dev).reshape(b * H, D)
            L = H
            saved = []
            for kind, st in self.stages:                        # ---------------- forward, keeping a1 / a2 per block
                if kind == "block":
                    e = B.linear(memb, st.emb_w, st.emb_b)                                                     # (b, co)
                    a1 = B.conv1d(cur, st.w1, st.b1, b, L, 1, st.pad)
                    h1 = B.groupnorm(a1, st.g1, st.be1, b, L, st.groups, act="mish", fa=e, fa_per_sample=True, film_mode=2)
                    a2 = B.conv1d(h1, st.w2, st.b2, b, L, 1, st.pad)
                    res = B.conv1d(cur, st.wr, st.br, b, L, 1, 0) if st.has_res else cur
                    cur = B.groupnorm(a2, st.g2, st.be2, b, L, st.groups, act="mish", residual=res)
                    saved.append((a1, a2, L))
                else:
                    cur = B.conv1d(cur, st.w, st.b, b, L, 2, 1)
                    saved.append((None, None, L))
                    L = (L - 1) // 2 + 1
            flat = cur.reshape(b, self.fc_in)                   # rows (b, l) x C  ==  [l][c] order of fc1_wx
            u = B.linear(emb, self.fc1_we, self.fc1_b)
            u = B.linear(flat, self.fc1_wx, None, residual=u)
            logp = B.linear(B.activation(u, "mish"), self.fc2_w, self.fc2_b)
            # ---------------- backward of logp.sum()
            ones = torch.ones_like(logp)
            du = B.linear(ones, self.fc2_w_t, None, gate=B.activation(u, "mish_grad"), rows_per_gate=1)
            grad = B.linear(du, self.fc1_wx_t, None).reshape(b * self.l_last, self.c_last)
            for (kind, st), (a1, a2, L) in zip(reversed(self.stages), reversed(saved)):
                if kind == "down" and L == 1:               # a single position: only the centre tap ever touched it
                    grad = B.conv1d(grad, st.even, None, b, 1, 1, 0, l_out=1)
                elif kind == "down":
                    lo = L // 2
                    c = st.even.shape[0]
                    full = torch.empty((b * L, c), device=dev, dtype=torch.float32)
                    view = full.view(b * lo, 2 * c)
                    B.conv1d(grad, st.even, None, b, lo, 1, 0, out=view[:, :c], l_out=lo)
                    B.conv1d(grad, st.odd, None, b, lo, 1, 0, out=view[:, c:], l_out=lo)
                    grad = full
                else:
                    da2 = B.groupnorm_backward(grad, a2, st.g2, st.be2, b, L, st.groups)
                    dh1 = B.conv1d(da2, st.w2_b, None, b, L, 1, st.pad)
                    da1 = B.groupnorm_backward(dh1, a1, st.g1, st.be1, b, L, st.groups)
                    dres = B.conv1d(grad, st.wr_b, None, b, L, 1, 0) if st.has_res else grad
                    grad = B.conv1d(da1, st.w1_b, None, b, L, 1, st.pad, residual=dres)
            return logp, grad.reshape(b, H, D)


_cache = weakref.WeakKeyDictionary()


def bound_for(net, dev) -> Optional[HalfJannerGrad]:
    sig = _signature(net)
    hit = _cache.get(net)
    if hit is None or hit[0] != sig:
        hit = (sig, HalfJannerGrad(net, dev))
        _cache[net] = hit
    return hit[1] if hit[1].ok else None


def gradients(classifier, x, noise, c):
    """Native (logp, grad) for CumRewClassifier-style classifiers over HalfJannerUNet1d, or None for the autograd path."""
    from ..classifier.rew_classifiers import CumRewClassifier
    from ..nn_classifier.half_jannerunet import HalfJannerUNet1d
    net = classifier.model_ema
    if type(classifier) is not CumRewClassifier or type(net) is not HalfJannerUNet1d or not x.is_cuda or \
            x.dtype != torch.float32 or x.dim() != 3:
        return None
    h = bound_for(net, x.device)
    if h is None:
        return None
    return h(x, noise, None)                                    # CumRewClassifier.logp ignores `c` (rew_classifiers.py:28-29)
