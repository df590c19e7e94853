"""Building blocks that appear inside the backbones on the sampling path.

* ``GroupNorm1d`` -- reference utils/building_blocks.py:60-76: ``num_groups = min(G, dim // min_cpg)``
  (SURVEY Q10), eps 1e-5, affine, applied to (b, C) or (b, C, L).
* ``Mlp`` -- reference utils/building_blocks.py:13-57 (``mlp.{i}.0`` Linear naming kept for checkpoints).
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F


class GroupNorm1d(nn.Module):
    def __init__(self, dim, num_groups=32, min_channels_per_group=4, eps=1e-5):
        super().__init__()
        self.num_groups = min(num_groups, dim // min_channels_per_group)
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def forward(self, x):
        y = F.group_norm(x.unsqueeze(2), self.num_groups, self.weight.to(x.dtype), self.bias.to(x.dtype), self.eps)
        return y.squeeze(2)


class Mlp(nn.Module):
    def __init__(self, in_dim: int, hidden_dims: List[int], out_dim: int,
                 activation: nn.Module = nn.ReLU(), out_activation: nn.Module = nn.Identity()):
        super().__init__()
        widths = [in_dim] + list(hidden_dims)
        hidden = [nn.Sequential(nn.Linear(a, b), activation) for a, b in zip(widths[:-1], widths[1:])]
        self.mlp = nn.Sequential(*hidden, nn.Linear(widths[-1], out_dim), out_activation)

    def _init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        return self.mlp(x)
