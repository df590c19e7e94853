"""SfBCUNet -- the MLP "U-Net" of SfBC (interface/checkpoint contract: reference nn_diffusion/sfbc_unet.py:9-82).

Residual blocks of Linears (no convolutions): ``block(x, c) = SiLU(L2(SiLU(L1 x) + Lc c)) + skip(x)``, a down path through
``hidden_dims``, a middle block, and an up path that concatenates the matching down activation.  The context ``c`` is
``t_layer(map_noise(t)) + condition``.  ``state_dict`` keys: ``t_layer.{0,2}``, ``{down,up}_blocks.{i}.{linear1.0,linear2.0,
linearc,skip}``, ``mid_block...``, ``out_layer``.  PyTorch executor (CPU, autograd, ROCm device through ATen).
"""
from typing import List, Optional

import torch
import torch.nn as nn

from .base_nn_diffusion import BaseNNDiffusion


class ResidualBlock(nn.Module):
    def __init__(self, in_dim: int, out_dim: int, emb_dim: int):
        super().__init__()
        self.linear1 = nn.Sequential(nn.Linear(in_dim, out_dim), nn.SiLU())
        self.linear2 = nn.Sequential(nn.Linear(out_dim, out_dim), nn.SiLU())
        self.linearc = nn.Linear(emb_dim, out_dim)
        self.skip = nn.Linear(in_dim, out_dim) if in_dim != out_dim else nn.Identity()

    def forward(self, x: torch.Tensor, c: torch.Tensor):
        return self.linear2(self.linear1(x) + self.linearc(c)) + self.skip(x)


class SfBCUNet(BaseNNDiffusion):
    def __init__(self, act_dim: int, emb_dim: int = 64, hidden_dims: List[int] = (512, 256, 128),
                 timestep_emb_type: str = "untrainable_fourier", timestep_emb_params: Optional[dict] = None):
        super().__init__(emb_dim, timestep_emb_type, timestep_emb_params)
        widths = [act_dim] + list(hidden_dims)
        self.t_layer = nn.Sequential(nn.Linear(emb_dim, emb_dim), nn.SiLU(), nn.Linear(emb_dim, emb_dim))
        self.down_blocks = nn.ModuleList([ResidualBlock(a, b, emb_dim) for a, b in zip(widths[:-1], widths[1:])])
        self.up_blocks = nn.ModuleList()
        top = widths[-1]
        self.mid_block = ResidualBlock(top, top, emb_dim)
        cur = top
        for skip_w, out_w in zip(reversed(widths[2:]), reversed(widths[1:-1])):     # concat with the matching down output
            self.up_blocks.append(ResidualBlock(cur + skip_w, out_w, emb_dim))
            cur = out_w
        self.out_layer = nn.Linear(cur, act_dim)

    def forward(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor] = None):
        """x (b, [horizon,] act_dim), noise (b,), condition (b, emb_dim)|None(=zeros) -> like x."""
        from ..engine import train
        if train.supports_sfbc(self, x, condition):
            return train.sfbc_forward(self, x, noise, condition)            # autograd on, ROCm device: loss() / update()
        c = self.t_layer(self.map_noise(noise))
        c = c + (condition if condition is not None else torch.zeros_like(c))
        kept = []
        for block in self.down_blocks:
            x = block(x, c)
            kept.append(x)
        x = self.mid_block(x, c)
        for block in self.up_blocks:
            x = block(torch.cat([x, kept.pop()], dim=-1), c)
        return self.out_layer(x)
