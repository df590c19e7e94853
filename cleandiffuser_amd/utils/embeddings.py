"""Timestep embeddings (the ``map_noise`` plug-ins of every backbone).

Registry contract: reference cleandiffuser/utils/utils.py:248-336 (names, ctor args, output
layout, the ``freqs`` buffer living in the checkpoint for the Fourier variants).

Engine note: the fused sampler never evaluates these per sample on the device.  The step
schedule is known before the loop starts, so ``sample()`` evaluates ``map_noise`` ONCE on the
(steps,) vector of timesteps (with the dtype the reference would use, which reproduces the
integer-timestep truncation quirk Q1) and hands the resulting (steps, emb_dim) table to the
kernel.
"""
import math

import numpy as np
import torch
import torch.nn as nn


def _outer(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    # `freqs` is cast to x's dtype first: for int64 timesteps this truncates every
    # frequency < 1 to 0 -- exactly what the reference does.
    return torch.einsum("...i,j->...ij", x, freqs.to(x.dtype))


def _geometric_freqs(dim: int, max_positions: int, endpoint: bool, device) -> torch.Tensor:
    half = dim // 2
    ramp = torch.arange(start=0, end=half, dtype=torch.float32, device=device)
    ramp = ramp / (half - (1 if endpoint else 0))
    return (1 / max_positions) ** ramp


class PositionalEmbedding(nn.Module):
    """[cos(t f_j), sin(t f_j)], f_j = max_positions^(-j/(dim/2)).  (b,) -> (b, dim)."""

    def __init__(self, dim: int, max_positions: int = 10000, endpoint: bool = False):
        super().__init__()
        self.dim, self.max_positions, self.endpoint = dim, max_positions, endpoint

    def forward(self, x):
        ang = x.ger(_geometric_freqs(self.dim, self.max_positions, self.endpoint, x.device).to(x.dtype))
        return torch.cat([ang.cos(), ang.sin()], dim=1)


class UntrainablePositionalEmbedding(PositionalEmbedding):
    """Same features, any leading shape (einsum instead of ger)."""

    def forward(self, x):
        ang = _outer(x, _geometric_freqs(self.dim, self.max_positions, self.endpoint, x.device))
        return torch.cat([ang.cos(), ang.sin()], dim=1)


class SinusoidalEmbedding(nn.Module):
    """Transformer token-position embedding, [sin, cos] order.  (...,) -> (..., dim)."""

    def __init__(self, dim: int):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        half = self.dim // 2
        decay = math.log(10000) / (half - 1)
        freqs = torch.exp(torch.arange(half, device=x.device) * -decay)
        ang = _outer(x, freqs)
        return torch.cat((ang.sin(), ang.cos()), dim=-1)


class FourierEmbedding(nn.Module):
    """Random Fourier features (dim/4 of them) followed by Linear-Mish-Linear."""

    def __init__(self, dim: int, scale=16):
        super().__init__()
        self.freqs = nn.Parameter(torch.randn(dim // 8) * scale, requires_grad=False)
        self.mlp = nn.Sequential(nn.Linear(dim // 4, dim), nn.Mish(), nn.Linear(dim, dim))

    def forward(self, x: torch.Tensor):
        ang = _outer(x, 2 * np.pi * self.freqs)
        feats = torch.cat([ang.cos(), ang.sin()], -1)
        if feats.is_cuda:
            from ..engine import heads                   # Linear-Mish-Linear on the library's own GEMM (gradient-free calls only)
            y = heads.try_sequential(self.mlp, feats)
            if y is not None:
                return y
        return self.mlp(feats)


class UntrainableFourierEmbedding(nn.Module):
    def __init__(self, dim: int, scale=16):
        super().__init__()
        self.freqs = nn.Parameter(torch.randn(dim // 2) * scale, requires_grad=False)

    def forward(self, x: torch.Tensor):
        ang = _outer(x, 2 * np.pi * self.freqs)
        return torch.cat([ang.cos(), ang.sin()], -1)


SUPPORTED_TIMESTEP_EMBEDDING = {
    "positional": PositionalEmbedding,
    "fourier": FourierEmbedding,
    "untrainable_fourier": UntrainableFourierEmbedding,
    "untrainable_positional": UntrainablePositionalEmbedding,
}
