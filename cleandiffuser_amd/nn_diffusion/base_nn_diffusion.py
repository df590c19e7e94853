"""Backbone plug-in base class (contract: reference nn_diffusion/base_nn_diffusion.py:9-42).

``forward(x:(b,horizon,in_dim), noise:(b,), condition:(b,emb_dim)|None) -> (b,horizon,in_dim)``.
``self.map_noise`` is built from the timestep-embedding registry so user backbones
(reference tutorials/4_*) keep working unchanged.
"""
from typing import Optional

import torch
import torch.nn as nn

from ..utils import SUPPORTED_TIMESTEP_EMBEDDING


class BaseNNDiffusion(nn.Module):
    def __init__(self, emb_dim: int, timestep_emb_type: str = "positional",
                 timestep_emb_params: Optional[dict] = None):
        assert timestep_emb_type in SUPPORTED_TIMESTEP_EMBEDDING.keys()
        super().__init__()
        self.map_noise = SUPPORTED_TIMESTEP_EMBEDDING[timestep_emb_type](emb_dim, **(timestep_emb_params or {}))

    def forward(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor] = None):
        raise NotImplementedError
