"""Small host helpers shared by the solver and backbone mirrors (reference utils.py:11-75,236-245,401-483)."""
import os
import random
from typing import Callable, Dict, Union

import numpy as np
import torch
import torch.nn as nn


def set_seed(seed: int):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def at_least_ndim(x, ndim: int, pad: int = 0):
    """Append (pad=0) or prepend (pad=1) singleton axes until x has `ndim` axes; scalars pass through."""
    if isinstance(x, (int, float)):
        return x
    if not isinstance(x, (np.ndarray, torch.Tensor)):
        raise ValueError(f"Unsupported type {type(x)}")
    missing = ndim - x.ndim
    if missing <= 0:
        return x
    shape = tuple(x.shape) + (1,) * missing if pad == 0 else (1,) * missing + tuple(x.shape)
    return x.reshape(shape)


def to_tensor(x, device=None):
    if isinstance(x, torch.Tensor):
        return x.to(device)
    if isinstance(x, (np.ndarray, list, tuple, int, float)):
        return torch.tensor(x, device=device)
    raise ValueError(f"Unsupported type {type(x)}")


def count_parameters(model: nn.Module):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def ema_update(model: nn.Module, model_ema: nn.Module, ema_rate: float):
    """p_ema <- rate * p_ema + (1 - rate) * p (reference diffusion/basic.py:83-86).  The update goes through the parameter
    itself (not ``.data``): in-place ops on ``.data`` do not bump ``Tensor._version``, which the native executors' packed-weight
    caches key on -- a stale cache would keep sampling from the first EMA weights.  The explicit epoch bump covers the caches
    for callers that still write through ``.data`` and then call ``invalidate_weights`` themselves."""
    first = next(model_ema.parameters(), None)
    if first is not None and first.is_cuda:          # ROCm device: ONE multi-tensor launch (csrc/cdx_optim.hip) instead of two per tensor
        from ..engine.optim import ema_update_native
        if ema_update_native(model, model_ema, ema_rate):
            invalidate_weights(model_ema)
            return
    with torch.no_grad():
        for p, p_ema in zip(model.parameters(), model_ema.parameters()):
            p_ema.mul_(ema_rate).add_(p.detach(), alpha=1 - ema_rate)
    invalidate_weights(model_ema)


def invalidate_weights(module: nn.Module):
    """Tell the native executors that `module`'s parameters changed behind autograd's back (``p.data`` writes,
    ``load_state_dict(assign=True)``...): every packed-weight / program cache keyed on a sub-module is rebuilt on next use."""
    for m in module.modules():
        m.__dict__["_cdx_epoch"] = m.__dict__.get("_cdx_epoch", 0) + 1


class _ModuleStateSwitch:
    """Context manager that flips a per-module/per-parameter flag and restores it on exit."""

    def __init__(self, modules):
        self.modules = list(modules)
        self._saved = {}


class FreezeModules(_ModuleStateSwitch):
    def __enter__(self):
        for m in self.modules:
            for p in m.parameters():
                self._saved[id(p)] = p.requires_grad
                p.requires_grad = False

    def __exit__(self, *exc):
        for m in self.modules:
            for p in m.parameters():
                p.requires_grad = self._saved[id(p)]


class UnfreezeModules(_ModuleStateSwitch):
    def __enter__(self):
        for m in self.modules:
            for p in m.parameters():
                self._saved[id(p)] = p.requires_grad
                p.requires_grad = True

    def __exit__(self, *exc):
        for m in self.modules:
            for p in m.parameters():
                p.requires_grad = self._saved[id(p)]


class EvalModules(_ModuleStateSwitch):
    def __enter__(self):
        for m in self.modules:
            self._saved[id(m)] = m.training
            m.eval()

    def __exit__(self, *exc):
        for m in self.modules:
            m.train(self._saved[id(m)])


class TrainModules(_ModuleStateSwitch):
    def __enter__(self):
        for m in self.modules:
            self._saved[id(m)] = m.training
            m.train()

    def __exit__(self, *exc):
        for m in self.modules:
            m.train(self._saved[id(m)])


def dict_apply(x: Dict[str, torch.Tensor], func: Callable[[torch.Tensor], torch.Tensor]):
    out = {}
    for k, v in x.items():
        out[k] = dict_apply(v, func) if isinstance(v, dict) else (None if v is None else func(v))
    return out


def loop_dataloader(dl):
    while True:
        yield from dl


TensorDict = Dict[str, Union["TensorDict", torch.Tensor]]


def param_to_module(param: str) -> str:
    """'a.b.weight' -> 'a.b' (owner module path of a parameter name)."""
    return param.rpartition(".")[0] if "." in param else param


def _human(n: int) -> str:
    return f"{n / 1e6:.2f} M" if n >= 1e6 else f"{n / 1e3:.2f} k"


def report_parameters(model, topk: int = 10) -> int:
    """Print the trainable-parameter total and the `topk` largest tensors with their owning module; return the total
    (what every reference pipeline prints after building its networks; reference utils/utils.py:355-376)."""
    sizes = {name: p.numel() for name, p in model.named_parameters() if p.requires_grad}
    total = sum(sizes.values())
    print(f"Total parameters: {_human(total)}")
    owners = dict(model.named_modules())
    ranked = sorted(sizes, key=lambda k: -sizes[k])
    for name in ranked[:topk]:
        print(" " * 8, f"{name:10}: {_human(sizes[name])} | {owners[name.rpartition('.')[0]]}")
    rest = ranked[topk:]
    print(" " * 8, f"... and {len(rest)} others accounting for {_human(sum(sizes[k] for k in rest))} parameters")
    return total


# Return normalisers of the Decision-Diffuser pipelines (discount 0.997), keyed by D4RL dataset name (reference utils/utils.py:379-395)
DD_RETURN_SCALE = {
    **{f"halfcheetah-{k}-v2": v for k, v in (("medium-expert", 3600), ("medium-replay", 1600), ("medium", 1700))},
    **{f"hopper-{k}-v2": v for k, v in (("medium-expert", 1200), ("medium-replay", 1000), ("medium", 1000))},
    **{f"walker2d-{k}-v2": v for k, v in (("medium-expert", 1600), ("medium-replay", 1300), ("medium", 1300))},
    "kitchen-partial-v0": 470, "kitchen-mixed-v0": 400,
    **{f"antmaze-{k}-v2": 100 for k in ("medium-play", "medium-diverse", "large-play", "large-diverse")},
}
