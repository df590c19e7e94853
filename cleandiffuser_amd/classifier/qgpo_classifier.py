"""QGPOClassifier (contract: reference classifier/qgpo_classifier.py:9-77): energy guidance of QGPO.  ``logp`` is the EMA energy
network f_phi(a_t, t, s); training is the in-support contrastive energy prediction loss
``-sum_k softmax(Q)_k log softmax_k f_phi(a_t^k, t, s)`` over K support actions per state."""
from typing import Dict

import torch
import torch.nn.functional as F

from .base import BaseClassifier


class QGPOClassifier(BaseClassifier):
    def loss(self, x: torch.Tensor, t: torch.Tensor, y: Dict[str, torch.Tensor]):
        """x (b, K, act_dim) noisy support actions, t (b,), y = {"soft_label": (b, K, 1), "obs": (b, obs_dim)}."""
        k = x.shape[1]
        energy = self.model(x, t.unsqueeze(1).repeat(1, k), y["obs"].unsqueeze(1).repeat(1, k, 1))
        loss = -(y["soft_label"] * F.log_softmax(energy, 1)).sum(1).mean()
        with torch.no_grad():
            stats = {"f_max": energy.max(1)[0].mean().item(), "f_mean": energy.mean().item(),
                     "f_min": energy.min(1)[0].mean().item()}
        return loss, stats

    def update(self, x: torch.Tensor, noise: torch.Tensor, y: Dict[str, torch.Tensor], update_ema: bool = True):
        loss, log = self.loss(x, noise, y)
        self.optim.zero_grad()
        loss.backward()
        grad_norm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), self.grad_clip_norm).item() \
            if isinstance(self.grad_clip_norm, float) else None
        self.optim.step()
        if update_ema:
            self.ema_update()
        log.update({"loss": loss.item(), "grad_norm": grad_norm})
        return log

    def logp(self, x: torch.Tensor, t: torch.Tensor, c: torch.Tensor):
        return self.model_ema(x, t, c)
