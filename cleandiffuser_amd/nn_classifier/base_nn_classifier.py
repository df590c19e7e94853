"""Classifier-network plug-in base (contract: reference nn_classifier/base_nn_classifier.py:9-33):
``forward(x, t, y) -> (b, 1)`` = log p(y | x, t) + C; owns ``self.map_noise`` like the diffusion backbones."""
from typing import Optional

import torch
import torch.nn as nn

from ..utils import SUPPORTED_TIMESTEP_EMBEDDING


class BaseNNClassifier(nn.Module):
    def __init__(self, emb_dim: int, timestep_emb_type: str = "positional", timestep_emb_params: Optional[dict] = None):
        assert timestep_emb_type in SUPPORTED_TIMESTEP_EMBEDDING.keys()
        super().__init__()
        self.map_noise = SUPPORTED_TIMESTEP_EMBEDDING[timestep_emb_type](emb_dim, **(timestep_emb_params or {}))

    def forward(self, x: torch.Tensor, t: torch.Tensor, y: torch.Tensor):
        raise NotImplementedError
