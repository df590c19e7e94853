"""MSEClassifier (contract: reference classifier/mse_classifier.py:10-29): ``logp(y | x, t) = -temperature * mean((f(x, t) - y)^2)``
with f the EMA network; trained by plain MSE regression of y."""
from typing import Optional

import torch
import torch.nn as nn

from .base import BaseClassifier


class MSEClassifier(BaseClassifier):
    def __init__(self, nn_classifier, temperature: float = 1.0, ema_rate: float = 0.995,
                 grad_clip_norm: Optional[float] = None, optim_params: Optional[dict] = None, device: str = "cpu"):
        super().__init__(nn_classifier, ema_rate, grad_clip_norm, optim_params, device)
        self.temperature = temperature

    def loss(self, x: torch.Tensor, noise: torch.Tensor, y: torch.Tensor):
        return nn.functional.mse_loss(self.model(x, noise), y)

    def logp(self, x: torch.Tensor, noise: torch.Tensor, c: torch.Tensor):
        return -self.temperature * ((self.model_ema(x, noise) - c) ** 2).mean(-1, keepdim=True)
