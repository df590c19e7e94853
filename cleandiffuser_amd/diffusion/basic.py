"""``DiffusionModel`` -- owner of the (diffusion, condition) networks, their EMA copy and the optimiser.

Interface contract: reference diffusion/basic.py:14-103 (constructor kwargs, ``.model`` / ``.model_ema``
``nn.ModuleDict{"diffusion","condition"}``, ``.optimizer``, ``.classifier``, ``.fix_mask[None]``,
``.loss_weight[None]``, ``train/eval/ema_update/save/load`` with the ``{"model","model_ema"}`` checkpoint).
"""
from copy import deepcopy
from typing import Optional, Union

import numpy as np
import torch
import torch.nn as nn

from ..nn_condition import BaseNNCondition, IdentityCondition
from ..nn_diffusion import BaseNNDiffusion
from ..utils import to_tensor


class DiffusionModel:
    def __init__(self, nn_diffusion: BaseNNDiffusion, nn_condition: Optional[BaseNNCondition] = None,
                 fix_mask: Union[list, np.ndarray, torch.Tensor] = None,
                 loss_weight: Union[list, np.ndarray, torch.Tensor] = None,
                 classifier=None, grad_clip_norm: Optional[float] = None, diffusion_steps: int = 1000,
                 ema_rate: float = 0.995, optim_params: Optional[dict] = None,
                 device: Union[torch.device, str] = "cpu"):
        self.device = device
        self.grad_clip_norm = grad_clip_norm
        self.diffusion_steps = diffusion_steps
        self.ema_rate = ema_rate

        nets = {"diffusion": nn_diffusion.to(device),
                "condition": (nn_condition if nn_condition is not None else IdentityCondition()).to(device)}
        self.model = nn.ModuleDict(nets)
        self.model_ema = deepcopy(self.model).requires_grad_(False)
        self.model.train()
        self.model_ema.eval()

        # a torch.optim.AdamW (subclass): identical on the CPU; on a ROCm device step() is one multi-tensor gfx950 kernel that also
        # clips, folds the EMA and zeroes the gradients (engine/optim.py, csrc/cdx_optim.hip)
        from ..engine.optim import FusedAdamW
        self.optimizer = FusedAdamW(self.model.parameters(), **(optim_params or {"lr": 2e-4, "weight_decay": 1e-5}))
        self.classifier = classifier

        self.fix_mask = 0. if fix_mask is None else to_tensor(fix_mask, device)[None, ]
        self.loss_weight = 1. if loss_weight is None else to_tensor(loss_weight, device)[None, ]

    # -- mode / EMA ---------------------------------------------------------------------------- #
    def _cached_plan(self, key, build):
        """Step plans depend only on (solver, grid, step counts): keep the last few so that a control loop calling sample() with the
        same settings re-derives nothing on the host (3 ms of scalar tensor math for 100 steps) and re-uploads nothing."""
        plans = self.__dict__.setdefault("_plans", {})
        if key not in plans:
            if len(plans) >= 16:
                plans.pop(next(iter(plans)))
            plans[key] = build()
        return plans[key]

    def train(self):
        self.model.train()
        if self.classifier is not None:
            self.classifier.model.train()

    def eval(self):
        self.model.eval()
        if self.classifier is not None:
            self.classifier.model.eval()

    def ema_update(self):
        from ..utils.misc import ema_update
        ema_update(self.model, self.model_ema, self.ema_rate)

    def _loss_backward(self, x0, condition=None, **kwargs):
        """``loss = self.loss(...); loss.backward()`` of every ``update()`` (reference diffusionsde.py:125-131, ddpm.py:98-104,
        newedm.py:178-184).  On a ROCm device, for the denoisers the native training path serves, the pair is ONE HIP-graph replay once
        the first step passed the capturability probe (engine/train.py:GraphedStep; CDX_TRAIN_GRAPH=0 keeps the eager pair)."""
        from ..engine import train
        g = train.graphed_step(self, x0, condition, kwargs)
        if g is None:
            loss = self.loss(x0, condition, **kwargs)
            with train.grads_in_place(self.model.parameters()):               # the library's nodes add parameter gradients straight into .grad
                loss.backward()
            return loss
        loss = g.replay(x0, condition)
        if hasattr(self.optimizer, "_gver"):
            # a replay writes gradients without moving their version counters: the ones the captured step writes ARE written (and only
            # those: a parameter this step does not reach keeps its "untouched since zeroed" mark, as under eager autograd)
            for p in g.written:
                self.optimizer._gver.pop(id(p), None)
        return loss

    def _apply_gradients(self, update_ema: bool = True, zero_grad: bool = True):
        """What every ``update()`` does after ``loss.backward()`` (reference diffusionsde.py:132-139): clip the global gradient
        norm, AdamW step, zero the gradients, EMA.  -> the clipped-from norm (tensor) or None, as the reference logs it.
        On a ROCm device with the library's optimiser this is <= 3 kernel launches for the whole model and no ATen optimiser launch."""
        from ..engine.optim import FusedAdamW
        opt = self.optimizer
        if isinstance(opt, FusedAdamW) and opt.native():
            opt.step(max_norm=self.grad_clip_norm or None, zero_grad=zero_grad,
                     ema=(self.model, self.model_ema, self.ema_rate) if update_ema else None)
            return opt.last_grad_norm if self.grad_clip_norm else None
        grad_norm = nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_clip_norm) if self.grad_clip_norm else None
        opt.step()
        if zero_grad:
            opt.zero_grad()
        if update_ema:
            self.ema_update()
        return grad_norm

    # -- abstract ------------------------------------------------------------------------------ #
    def update(self, x0, condition=None, update_ema=True, **kwargs):
        raise NotImplementedError

    def sample(self, *args, **kwargs):
        raise NotImplementedError

    # -- checkpoints --------------------------------------------------------------------------- #
    def save(self, path: str):
        torch.save({"model": self.model.state_dict(), "model_ema": self.model_ema.state_dict()}, path)

    def load(self, path: str):
        ckpt = torch.load(path, map_location=self.device)
        self.model.load_state_dict(ckpt["model"])
        self.model_ema.load_state_dict(ckpt["model_ema"])
