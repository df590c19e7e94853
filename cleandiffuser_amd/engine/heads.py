"""Row-wise heads evaluated on freshly sampled tensors: inverse dynamics (reference invdynamic/mlp.py:72,
called at pipelines/dd_d4rl_antmaze.py:142-143), candidate critics (utils/building_blocks.py:111-147, utils/iql.py:7-37,
pipelines/idql_d4rl_mujoco.py:183-194).  Every one of them is an ``nn.Sequential`` of Linear / LayerNorm / activation, so it is
compiled once into a list of launches on the caller's stream:

    Linear [+ activation]          -> one ``cdx_gemm_f32`` (bias + activation in the epilogue)
    LayerNorm [+ activation]       -> one ``cdx_groupnorm_f32`` with L = 1, G = 1 (per-row statistics over C, affine, activation)
    lone activation                -> ``cdx_act_f32``

No host synchronisation and no D2H: the heads consume ``sample()``'s output where it lies in HBM.  Anything else in the chain
(a module this table does not know, active dropout, autograd, non-fp32, CPU tensors) returns None and the stock modules run.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

_ACT_OF = {nn.ReLU: "relu", nn.Tanh: "tanh", nn.Mish: "mish", nn.SiLU: "silu"}


def _act_name(m: nn.Module) -> Optional[str]:
    for cls, name in _ACT_OF.items():
        if type(m) is cls:
            return name
    if type(m) is nn.GELU:
        return "gelu_tanh" if m.approximate == "tanh" else "gelu"
    if type(m) is nn.LeakyReLU and abs(m.negative_slope - 0.01) < 1e-12:
        return "leaky"
    return None


def _flatten(seq: nn.Module) -> List[nn.Module]:
    out = []
    for m in seq.children() if isinstance(seq, nn.Sequential) else [seq]:
        out += _flatten(m) if isinstance(m, nn.Sequential) else [m]
    return out


def compile_chain(seq: nn.Module) -> Optional[List[Tuple]]:
    """[('linear', Linear, act) | ('norm', LayerNorm, act) | ('act', name)] or None when a module is not understood."""
    mods = [m for m in _flatten(seq)
            if not isinstance(m, nn.Identity) and not (isinstance(m, nn.Dropout) and (not m.training or m.p == 0.0))]
    ops, i = [], 0
    while i < len(mods):
        m = mods[i]
        nxt = _act_name(mods[i + 1]) if i + 1 < len(mods) else None
        if type(m) is nn.Linear:
            ops.append(("linear", m, nxt or "none"))
        elif type(m) is nn.LayerNorm and m.elementwise_affine and len(m.normalized_shape) == 1 and m.bias is not None:
            ops.append(("norm", m, nxt or "none"))
        elif _act_name(m) is not None:
            ops.append(("act", _act_name(m)))
            nxt = None
        else:
            return None
        i += 2 if nxt else 1
    return ops


def native_ok(x: torch.Tensor, params) -> bool:
    if not x.is_cuda or x.dtype != torch.float32:
        return False
    params = list(params)
    if any(p.dtype != torch.float32 or p.device != x.device for p in params):
        return False                                   # mixed precision / split placement: whatever PyTorch makes of it
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
        return False
    return True


def run_chain(ops, x: torch.Tensor) -> torch.Tensor:
    from . import blocks
    lead = x.shape[:-1]
    h = x.reshape(-1, x.shape[-1])
    if h.stride(-1) != 1 or (h.shape[0] > 1 and h.stride(0) < h.shape[1]):
        h = h.contiguous()
    for op in ops:
        if op[0] == "linear":
            lin = op[1]
            h = blocks.linear(h, lin.weight, lin.bias, act=op[2])
        elif op[0] == "norm":
            ln = op[1]
            h = blocks.groupnorm(h, ln.weight, ln.bias, batch=h.shape[0], length=1, groups=1, act=op[2], eps=ln.eps)
        else:
            h = blocks.activation(h, op[1])
    return h.reshape(*lead, h.shape[-1])


def try_sequential(seq: nn.Module, x: torch.Tensor) -> Optional[torch.Tensor]:
    """``seq(x)`` through the HIP library, or None for the PyTorch path.  The compiled chain is cached on the module and rebuilt
    when train()/eval() flips (dropout) or a sub-module is replaced -- weights are read in place, so optimiser steps and
    ``load_state_dict`` need no invalidation."""
    if not native_ok(x, seq.parameters()):
        return None
    if x.numel() == 0:
        return None
    key = (bool(seq.training), tuple(id(m) for m in seq.modules()))        # train/eval flips and swapped sub-modules recompile
    cached = getattr(seq, "_cdx_chain", None)
    if cached is None or cached[0] != key:
        cached = (key, compile_chain(seq))
        object.__setattr__(seq, "_cdx_chain", cached)
    if cached[1] is None:
        return None
    return run_chain(cached[1], x)
