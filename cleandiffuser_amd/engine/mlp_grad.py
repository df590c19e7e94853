"""Classifier guidance without autograd for the MLP classifiers: ``(logp, d logp.sum() / d x)`` of

* ``MSEClassifier(MLPNNClassifier)``   -- logp = -temperature * mean((mlp([x | map_noise(t)]) - c)^2)   (reference
  classifier/mse_classifier.py:27-29 over nn_classifier/mlp.py:10-22),
* ``QGPOClassifier(QGPONNClassifier)`` -- logp = 10 tanh(mlp([obs_proj(s) | act_proj(a_t) | map_noise(t)]) / 10)   (reference
  classifier/qgpo_classifier.py:63-77 over nn_classifier/mlp.py:25-55; QGPO's energy guidance, evaluated at EVERY denoising step),

by an explicit forward + backward through the ``Mlp`` chain on the library's kernels: every Linear forward is ``cdx_gemm_f32`` (the
pre-activation is kept), every activation ``cdx_act_f32``; the backward multiplies by ``act'(pre)`` (``cdx_act_bwd_f32``) and applies
the TRANSPOSED weights with the same GEMM.  What it replaces: ``torch.autograd.grad(logp.sum(), x)`` in ``BaseClassifier.gradients``
(reference classifier/base.py:74-79).  Only the input gradient is produced.  Anything this does not recognise returns None and the
caller differentiates with autograd as the reference does.
"""
import weakref
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from . import blocks as B
from .heads import _act_name
from .runtime import _f32c, _signature


def _mlp_layers(mlp) -> Optional[List[Tuple[nn.Linear, str]]]:
    """utils.Mlp -> [(Linear, activation name | 'none')] in forward order, or None when a layer is not understood."""
    seq = getattr(mlp, "mlp", None)
    if not isinstance(seq, nn.Sequential) or len(seq) < 2:
        return None
    layers = []
    for m in list(seq)[:-2]:
        if not (isinstance(m, nn.Sequential) and len(m) == 2 and type(m[0]) is nn.Linear):
            return None
        act = _act_name(m[1])
        if act is None or act == "gelu_tanh":
            return None
        layers.append((m[0], act))
    last, out_act = seq[-2], seq[-1]
    if type(last) is not nn.Linear:
        return None
    if isinstance(out_act, nn.Identity):
        layers.append((last, "none"))
    else:
        act = _act_name(out_act)
        if act is None or act == "gelu_tanh":
            return None
        layers.append((last, act))
    return layers


class _Chain:
    """Forward with saved pre-activations + input-gradient backward of one Mlp; the transposed weights are made once per weight version."""

    def __init__(self, layers, dev, grad_cols: slice):
        self.layers = layers
        self.wt = []
        for i, (lin, _) in enumerate(layers):
            w = lin.weight.detach()
            if i == 0:
                w = w[:, grad_cols]                       # only the columns of the input the caller differentiates
            self.wt.append(_f32c(w.t(), dev).contiguous())

    def forward(self, h: torch.Tensor):
        pres = []
        for lin, act in self.layers:
            pre = B.linear(h, lin.weight, lin.bias)
            pres.append(pre)
            h = pre if act == "none" else B.activation(pre, act)
        return h, pres

    def backward(self, g: torch.Tensor, pres) -> torch.Tensor:
        for i in range(len(self.layers) - 1, -1, -1):
            act = self.layers[i][1]
            if act != "none":
                g = B.activation_backward(pres[i], g, act)
            g = B.linear(g, self.wt[i])
        return g


_cache = weakref.WeakKeyDictionary()


def _bound(net, dev, make):
    sig = _signature(net)
    hit = _cache.get(net)
    if hit is None or hit[0] != sig:
        hit = (sig, make())
        _cache[net] = hit
    return hit[1]


def gradients(classifier, x: torch.Tensor, noise: torch.Tensor, c):
    """Native (logp, grad) for the two MLP classifier wrappers, or None for the autograd path."""
    from ..classifier.mse_classifier import MSEClassifier
    from ..classifier.qgpo_classifier import QGPOClassifier
    from ..nn_classifier.mlp import MLPNNClassifier, QGPONNClassifier
    net = classifier.model_ema
    if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or c is None or net.training:
        return None
    dev = x.device
    with torch.no_grad():
        if type(classifier) is MSEClassifier and type(net) is MLPNNClassifier:
            layers = _mlp_layers(net.mlp)
            if layers is None or not isinstance(c, torch.Tensor) or c.dim() != 2:
                return None
            xd = x.shape[1]
            chain = _bound(net, dev, lambda: _Chain(layers, dev, slice(0, xd)))
            feats = torch.cat([_f32c(x, dev), _f32c(net.map_noise(noise), dev)], dim=-1).contiguous()
            pred, pres = chain.forward(feats)
            diff = pred - _f32c(c, dev)
            temp = float(classifier.temperature)
            logp = -temp * (diff ** 2).mean(-1, keepdim=True)
            g = ((-2.0 * temp / diff.shape[1]) * diff).contiguous()
            return logp, chain.backward(g, pres)
        if type(classifier) is QGPOClassifier and type(net) is QGPONNClassifier:
            layers = _mlp_layers(net.mlp)
            if layers is None or not isinstance(c, torch.Tensor) or c.dim() != 2 or layers[-1][1] != "none":
                return None
            e = net.act_proj.out_features
            # gradient w.r.t. the action projection's OUTPUT (columns [E, 2E) of the features), then through act_proj's weights
            chain = _bound(net, dev, lambda: (_Chain(layers, dev, slice(e, 2 * e)), _f32c(net.act_proj.weight.detach().t(), dev).contiguous()))
            chain, proj_t = chain
            feats = torch.cat([B.linear(_f32c(c, dev), net.obs_proj.weight, net.obs_proj.bias),
                               B.linear(_f32c(x, dev), net.act_proj.weight, net.act_proj.bias),
                               _f32c(net.map_noise(noise), dev)], dim=-1).contiguous()
            o, pres = chain.forward(feats)
            logp = torch.tanh(o / 10) * 10                                   # (b, 1): the squash of nn_classifier/mlp.py:55
            g = B.activation_backward(o, torch.ones_like(o), "tanh", param=10.0)
            g = chain.backward(g, pres)
            return logp, B.linear(g, proj_t)
    return None
