"""Classifier-guidance plug-in (contract: reference classifier/base.py:9-90): owns ``model`` / ``model_ema`` of a
``BaseNNClassifier``, ``logp(x, noise, c) -> (b, 1)``, ``gradients`` = autograd of ``logp.sum()`` w.r.t. ``x``,
Adam optimiser, EMA, ``save/load`` with the ``{"model","model_ema"}`` checkpoint."""
from copy import deepcopy
from typing import Optional

import torch


class BaseClassifier:
    def __init__(self, nn_classifier, ema_rate: float = 0.995, grad_clip_norm: Optional[float] = None,
                 optim_params: Optional[dict] = None, device: str = "cpu"):
        self.device = device
        self.ema_rate, self.grad_clip_norm = ema_rate, grad_clip_norm
        self.model = nn_classifier.to(device)
        self.model_ema = deepcopy(self.model).eval()
        self.optim = torch.optim.Adam(self.model.parameters(), **(optim_params or {"lr": 2e-4, "weight_decay": 1e-4}))

    def eval(self):
        self.model.eval()
        self.model_ema.eval()

    def train(self):
        self.model.train()

    def ema_update(self):
        with torch.no_grad():
            for p, p_ema in zip(self.model.parameters(), self.model_ema.parameters()):
                p_ema.data.mul_(self.ema_rate).add_(p.data, alpha=1. - self.ema_rate)

    def loss(self, x: torch.Tensor, noise: torch.Tensor, y: torch.Tensor):
        raise NotImplementedError

    def update(self, x: torch.Tensor, noise: torch.Tensor, y: torch.Tensor, update_ema: bool = True):
        loss = self.loss(x, noise, y)
        self.optim.zero_grad()
        loss.backward()
        grad_norm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_clip_norm).item() \
            if isinstance(self.grad_clip_norm, float) else None
        self.optim.step()
        if update_ema:
            self.ema_update()
        return {"loss": loss.item(), "grad_norm": grad_norm}

    def logp(self, x: torch.Tensor, noise: torch.Tensor, c: torch.Tensor):
        raise NotImplementedError

    def gradients(self, x: torch.Tensor, noise: torch.Tensor, c: torch.Tensor):
        if x.is_cuda:                         # explicit forward + backward kernels (engine/classifier_grad.py); None -> autograd
            from ..engine import classifier_grad
            native = classifier_grad.gradients(self, x, noise, c)
            if native is not None:
                return native
        x.requires_grad_()
        logp = self.logp(x, noise, c)
        grad = torch.autograd.grad([logp.sum()], [x])[0]
        x.detach()
        return logp.detach(), grad.detach()

    def save(self, path):
        torch.save({"model": self.model.state_dict(), "model_ema": self.model_ema.state_dict()}, path)

    def load(self, path):
        ckpt = torch.load(path, map_location=self.device)
        self.model.load_state_dict(ckpt["model"])
        self.model_ema.load_state_dict(ckpt["model_ema"])
