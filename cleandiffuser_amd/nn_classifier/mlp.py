"""MLP classifier backbones (contract: reference nn_classifier/mlp.py:10-55).

``MLPNNClassifier``: ``Mlp([x | map_noise(t)])`` -- a plain regressor / logit head on noisy inputs (``mlp.mlp.{i}.0`` keys).
``QGPONNClassifier``: QGPO's energy model f_phi(a_t, t | s): observation and action projected to ``emb_dim`` each, concatenated
with the time embedding, SiLU MLP to one scalar, squashed to (-10, 10) by ``10 tanh(out / 10)``.
"""
from typing import List

import torch
import torch.nn as nn

from ..utils import Mlp
from .base_nn_classifier import BaseNNClassifier


class MLPNNClassifier(BaseNNClassifier):
    def __init__(self, x_dim: int, out_dim: int, emb_dim: int, hidden_dims: List[int], activation: nn.Module = nn.ReLU(),
                 out_activation: nn.Module = nn.Identity(), timestep_emb_type: str = "positional"):
        super().__init__(emb_dim, timestep_emb_type)
        self.mlp = Mlp(x_dim + emb_dim, hidden_dims, out_dim, activation, out_activation)

    def forward(self, x: torch.Tensor, t: torch.Tensor, y: torch.Tensor = None):
        return self.mlp(torch.cat([x, self.map_noise(t)], dim=-1))


class QGPONNClassifier(BaseNNClassifier):
    def __init__(self, obs_dim: int, act_dim: int, emb_dim: int, hidden_dims: List[int],
                 timestep_emb_type: str = "positional"):
        super().__init__(emb_dim, timestep_emb_type)
        self.obs_proj = nn.Linear(obs_dim, emb_dim)
        self.act_proj = nn.Linear(act_dim, emb_dim)
        self.mlp = Mlp(3 * emb_dim, hidden_dims, 1, nn.SiLU())

    def forward(self, x: torch.Tensor, t: torch.Tensor, y: torch.Tensor):
        """x noisy actions (b, act_dim), t (b,), y observations (b, obs_dim) -> energy (b, 1) in (-10, 10)."""
        feats = torch.cat([self.obs_proj(y), self.act_proj(x), self.map_noise(t)], dim=-1)
        return torch.tanh(self.mlp(feats) / 10) * 10
