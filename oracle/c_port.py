"""ctypes front end of oracle/c/janner_oracle.c (plain-C restatement).  TEST INFRASTRUCTURE (oracle/__init__.py)."""
import ctypes
import os

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "_build", "libcdx_oracle.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            import sys
            sys.path.insert(0, os.path.dirname(_HERE))
            import __graft_entry__ as g
            g.build_oracle_c()
        _lib = ctypes.CDLL(LIB)
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def flat_params(net) -> np.ndarray:
    """One fp32 buffer in EXECUTION order (map_emb, downs, mid_block1, mid_block2, ups, final_conv; inside a group the
    state_dict order) -- the C walker consumes it with a cursor.  (state_dict itself lists ``ups`` before the mid blocks.)"""
    sd = net.state_dict()
    out = []
    for group in ("map_emb.", "downs.", "mid_block1.", "mid_block2.", "ups.", "final_conv."):
        out += [v.detach().cpu().numpy().astype(np.float32).ravel() for k, v in sd.items() if k.startswith(group)]
    assert sum(a.size for a in out) == sum(v.numel() for k, v in sd.items() if not k.startswith("map_noise"))
    return np.concatenate(out)


def janner_sample(net, x_init, prior, fix_mask, noise, temb, plan, predict_noise):
    """Run the C oracle's full loop.  `plan` = cleandiffuser_amd.engine.plan.SamplePlan (scalars only)."""
    lib = load()
    b, h, d = x_init.shape
    steps = np.zeros((len(plan.steps), 12), np.float32)
    k = 0
    for i, st in enumerate(plan.steps):
        nidx = -1
        if st.noise:
            nidx, k = k, k + 1
        steps[i] = [st.kind, st.vsel, nidx, int(st.push), st.alpha, st.sigma, *st.k, 0.0]
    x = np.ascontiguousarray(x_init, np.float32).copy()
    params = flat_params(net)
    mult = (ctypes.c_int * len(net_dim_mult(net)))(*net_dim_mult(net))
    prior = np.ascontiguousarray(prior, np.float32)
    fm = None if fix_mask is None else np.ascontiguousarray(fix_mask, np.float32)
    nz = None if noise is None else np.ascontiguousarray(noise, np.float32)
    temb = np.ascontiguousarray(temb, np.float32)
    lib.cdx_oracle_janner_sample(_p(params), _p(x), _p(prior), _p(fm), _p(nz), _p(temb), _p(steps),
                                 ctypes.c_int(len(plan.steps)), ctypes.c_int(int(predict_noise)), ctypes.c_int(b),
                                 ctypes.c_int(h), ctypes.c_int(d), ctypes.c_int(net.model_dim),
                                 ctypes.c_int(net.emb_dim), ctypes.c_int(net.kernel_size),
                                 ctypes.c_int(len(net.downs)), mult)
    return x


def net_dim_mult(net):
    widths = [net.model_dim] + [lvl[0].conv1[0].out_channels for lvl in net.downs]
    return [widths[i + 1] // widths[i] for i in range(len(widths) - 1)]
