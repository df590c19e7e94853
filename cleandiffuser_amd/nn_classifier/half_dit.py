"""HalfDiT1d -- DiT1d trunk as a trajectory classifier (contract: reference nn_classifier/half_dit.py:9-76): the final layer
maps tokens to ``d_model // 2`` features (zero-initialised like every DiT head), tokens are mean-pooled and a
LayerNorm-SiLU-Linear x2 head produces ``out_dim`` values.  The trunk is ``nn_diffusion.DiT1d``: with autograd on, on a ROCm device
-- which is how ``BaseClassifier.gradients`` differentiates ``logp`` (reference classifier/base.py:74-79) -- its Linear / LayerNorm +
modulate / attention nodes run forward and backward on the library's kernels (engine/train.py:dit_forward, round 5); the pooled head
(a few (batch, d_model / 2) rows) stays ATen.  Without autograd the class is the PyTorch executor (the sampling dispatch checks the
exact type DiT1d)."""
from typing import Optional

import torch
import torch.nn as nn

from ..nn_diffusion.dit import DiT1d, FinalLayer1d


class HalfDiT1d(DiT1d):
    def __init__(self, in_dim: int, out_dim: int, emb_dim: int, d_model: int = 384, n_heads: int = 6, depth: int = 12,
                 dropout: float = 0.0, timestep_emb_type: str = "positional"):
        super().__init__(in_dim, emb_dim, d_model, n_heads, depth, dropout, timestep_emb_type)
        self.final_layer = FinalLayer1d(d_model, d_model // 2)
        for lin in (self.final_layer.adaLN_modulation[-1], self.final_layer.linear):
            nn.init.constant_(lin.weight, 0)
            nn.init.constant_(lin.bias, 0)
        self.proj = nn.Sequential(nn.LayerNorm(d_model // 2), nn.SiLU(), nn.Linear(d_model // 2, d_model // 4),
                                  nn.LayerNorm(d_model // 4), nn.SiLU(), nn.Linear(d_model // 4, out_dim))

    def forward(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor] = None):
        """x (b, horizon, in_dim), noise (b,), condition (b, emb_dim)|None -> (b, out_dim)."""
        return self.proj(super().forward(x, noise, condition).mean(1))
