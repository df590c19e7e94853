"""Low-dimensional condition encoders (reference nn_condition/mlp.py:10-92, pearce_obs_condition.py:10-50,
positional.py:8-54).  Parameter names (``affine``, ``mlp.mlp.{i}.0``, ``mlp.{0,2}``, ``freqs``) match the
reference checkpoints.

Execution: the Linear / activation chains run through engine/heads.py (``cdx_gemm_f32`` with bias + activation in the epilogue)
on a ROCm device when no gradient is needed -- i.e. inside ``sample()`` -- and as stock modules otherwise (training, CPU)."""
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from ..utils import Mlp
from .base_nn_condition import IdentityCondition


def _rows(net: nn.Module, x: torch.Tensor) -> torch.Tensor:
    from ..engine import heads
    y = heads.try_sequential(net.mlp if isinstance(net, Mlp) else net, x)
    return net(x) if y is None else y


class LinearCondition(IdentityCondition):
    def __init__(self, in_dim: int, out_dim: int, dropout: float = 0.25):
        super().__init__(dropout)
        self.affine = nn.Linear(in_dim, out_dim)

    def forward(self, condition: torch.Tensor, mask: torch.Tensor = None):
        m = self._mask(mask, condition.shape[0], condition.device, condition.dim())
        return _rows(self.affine, condition) * m


class MLPCondition(IdentityCondition):
    def __init__(self, in_dim: int, out_dim: int, hidden_dims: List[int], act=nn.LeakyReLU(), dropout: float = 0.25):
        super().__init__(dropout)
        self.mlp = Mlp(in_dim, [hidden_dims] if isinstance(hidden_dims, int) else hidden_dims, out_dim, act)

    def forward(self, condition: torch.Tensor, mask: torch.Tensor = None):
        m = self._mask(mask, condition.shape[0], condition.device, condition.dim())
        return _rows(self.mlp, condition) * m


class MLPSieveObsCondition(IdentityCondition):
    """(b, history, o_dim) -> per-frame MLP -> flatten -> (b, history*emb_dim)."""

    def __init__(self, o_dim: int, emb_dim: int = 128, hidden_dim: int = 512, dropout: float = 0.25):
        super().__init__(dropout)
        self.mlp = Mlp(o_dim, [hidden_dim], emb_dim, nn.LeakyReLU())

    def forward(self, obs: torch.Tensor, mask: torch.Tensor = None):
        m = self._mask(mask, obs.shape[0], obs.device, 2)
        return torch.flatten(_rows(self.mlp, obs), 1) * m


class PearceObsCondition(IdentityCondition):
    """DiffusionBC frame encoder: shared 2-layer MLP per frame, optional flatten."""

    def __init__(self, obs_dim: int, emb_dim: int = 128, flatten: bool = False, dropout: float = 0.25):
        super().__init__(dropout)
        self.mlp = nn.Sequential(nn.Linear(obs_dim, emb_dim), nn.LeakyReLU(), nn.Linear(emb_dim, emb_dim))
        self.flatten = flatten

    def forward(self, obs: torch.Tensor, mask: Optional[torch.Tensor] = None):
        m = self._mask(mask, obs.shape[0], obs.device, 2 if self.flatten else 3)
        e = _rows(self.mlp, obs)
        return (torch.flatten(e, 1) if self.flatten else e) * m


class FourierCondition(MLPCondition):
    """(b,1) scalar -> random Fourier features -> MLP."""

    def __init__(self, out_dim, hidden_dim, scale=16, dropout=0.25):
        super().__init__(hidden_dim, out_dim, hidden_dim, nn.Mish(), dropout)
        self.register_buffer("freqs", torch.randn(hidden_dim // 2) * scale)

    def forward(self, condition: torch.Tensor, mask: torch.Tensor = None):
        ang = condition.squeeze(-1).ger((2 * np.pi * self.freqs).to(condition.dtype))
        return super().forward(torch.cat([ang.cos(), ang.sin()], -1), mask)


class PositionalCondition(MLPCondition):
    """(b,) scalar -> geometric positional features -> MLP."""

    def __init__(self, out_dim, hidden_dim, dropout=0.25, max_positions: int = 10000, endpoint: bool = False):
        super().__init__(hidden_dim, out_dim, hidden_dim, nn.Mish(), dropout)
        self.max_positions, self.endpoint, self.dim = max_positions, endpoint, out_dim

    def forward(self, condition: torch.Tensor, mask: torch.Tensor = None):
        half = self.dim // 2
        ramp = torch.arange(start=0, end=half, dtype=torch.float32, device=condition.device)
        freqs = (1 / self.max_positions) ** (ramp / (half - (1 if self.endpoint else 0)))
        ang = condition.ger(freqs.to(condition.dtype))
        return super().forward(torch.cat([ang.cos(), ang.sin()], dim=1), mask)
