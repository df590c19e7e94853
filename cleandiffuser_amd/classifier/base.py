"""Classifier-guidance plug-in (contract: reference classifier/base.py:9-90): owns ``model`` / ``model_ema`` of a
``BaseNNClassifier``, ``logp(x, noise, c) -> (b, 1)``, ``gradients(x, noise, c) -> (logp, d logp.sum() / dx)``, an Adam optimiser
(default lr 2e-4, weight decay 1e-4), an EMA copy, and ``save/load`` of the ``{"model", "model_ema"}`` checkpoint.

``gradients`` on a ROCm device is served by explicit forward + backward kernels when the network is one the engine knows
(engine/classifier_grad.py: HalfJannerUNet1d; engine/mlp_grad.py: MSEClassifier / QGPOClassifier over the MLP networks); everything else differentiates ``logp`` with autograd as the reference does.
"""
from copy import deepcopy
from typing import Optional

import torch

_DEFAULT_ADAM = {"lr": 2e-4, "weight_decay": 1e-4}


class BaseClassifier:
    def __init__(self, nn_classifier, ema_rate: float = 0.995, grad_clip_norm: Optional[float] = None,
                 optim_params: Optional[dict] = None, device: str = "cpu"):
        self.device = device
        self.ema_rate = ema_rate
        self.grad_clip_norm = grad_clip_norm
        self.model = nn_classifier.to(device)
        self.model_ema = deepcopy(self.model).eval()
        self.optim = torch.optim.Adam(self.model.parameters(), **(_DEFAULT_ADAM if optim_params is None else optim_params))

    # ---- what a concrete classifier defines ----
    def loss(self, x: torch.Tensor, noise: torch.Tensor, y: torch.Tensor):
        raise NotImplementedError

    def logp(self, x: torch.Tensor, noise: torch.Tensor, c: torch.Tensor):
        """log p(c | x_t, noise level): x (b, *x_shape), noise (b,), c (b, *c_shape) -> (b, 1)."""
        raise NotImplementedError

    # ---- guidance signal ----
    def gradients(self, x: torch.Tensor, noise: torch.Tensor, c: torch.Tensor):
        if x.is_cuda:
            from ..engine import classifier_grad
            native = classifier_grad.gradients(self, x, noise, c)        # None: not a network with hand-written backward kernels
            if native is None:
                from ..engine import mlp_grad                            # the MLP / QGPO energy classifiers: explicit GEMM backward
                native = mlp_grad.gradients(self, x, noise, c)
            if native is not None:
                return native
        x.requires_grad_()
        logp = self.logp(x, noise, c)
        (grad,) = torch.autograd.grad([logp.sum()], [x])
        x.detach()                                                       # (no-op, as in the reference: x keeps requires_grad)
        return logp.detach(), grad.detach()

    # ---- training ----
    def update(self, x: torch.Tensor, noise: torch.Tensor, y: torch.Tensor, update_ema: bool = True):
        loss = self.loss(x, noise, y)
        self.optim.zero_grad()
        loss.backward()
        grad_norm = None
        if isinstance(self.grad_clip_norm, float):
            grad_norm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_clip_norm).item()
        self.optim.step()
        if update_ema:
            self.ema_update()
        return {"loss": loss.item(), "grad_norm": grad_norm}

    def ema_update(self):
        from ..utils.misc import ema_update
        ema_update(self.model, self.model_ema, self.ema_rate)

    def train(self):
        self.model.train()

    def eval(self):
        for m in (self.model, self.model_ema):
            m.eval()

    # ---- checkpoints ----
    def save(self, path):
        torch.save({name: getattr(self, name).state_dict() for name in ("model", "model_ema")}, path)

    def load(self, path):
        ckpt = torch.load(path, map_location=self.device)
        for name in ("model", "model_ema"):
            getattr(self, name).load_state_dict(ckpt[name])
