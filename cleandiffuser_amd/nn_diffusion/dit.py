"""DiT1d / DiT1Ref -- adaLN-Zero transformer over trajectory tokens (Decision Diffuser & friends).
Interface/checkpoint contract: reference nn_diffusion/dit.py:10-180 (``x_proj``, ``map_emb.{0,2}``,
``blocks.{i}.{attn.in_proj_weight,attn.out_proj,mlp.{0,3},adaLN_modulation.1}``, ``final_layer``).

Reference quirk kept on purpose: the block overwrites ``x`` with the *modulated LayerNorm output* before
the attention residual, i.e. ``x <- mod(LN(x)); x <- x + gate * attn(x)`` -- not the textbook DiT residual.

Execution: on a ROCm device without autograd, ``DiT1d.forward`` and every ``sample()`` over it run through
``engine/bigbatch.py`` -> ``cdx_dit1d_run`` (tiled fp32-MFMA GEMMs with fused epilogues, LayerNorm+modulate, small-T
attention, csrc/cdx_gemm.hip + cdx_bigbatch.hip); the module code below is the CPU / autograd executor.
"""
from typing import Optional

import torch
import torch.nn as nn

from ..utils import SinusoidalEmbedding
from .base_nn_diffusion import BaseNNDiffusion


def modulate(x, shift, scale):
    return x * (1 + scale.unsqueeze(1)) + shift.unsqueeze(1)


class DiTBlock(nn.Module):
    def __init__(self, hidden_size: int, n_heads: int, dropout: float = 0.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.attn = nn.MultiheadAttention(hidden_size, n_heads, dropout, batch_first=True)
        self.norm2 = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.mlp = nn.Sequential(nn.Linear(hidden_size, hidden_size * 4), nn.GELU(approximate="tanh"),
                                 nn.Dropout(dropout), nn.Linear(hidden_size * 4, hidden_size))
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, hidden_size * 6))

    def forward(self, x: torch.Tensor, t: torch.Tensor):
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = self.adaLN_modulation(t).chunk(6, dim=1)
        x = modulate(self.norm1(x), sh_a, sc_a)
        x = x + g_a.unsqueeze(1) * self.attn(x, x, x)[0]
        return x + g_m.unsqueeze(1) * self.mlp(modulate(self.norm2(x), sh_m, sc_m))


class FinalLayer1d(nn.Module):
    def __init__(self, hidden_size: int, out_dim: int):
        super().__init__()
        self.norm_final = nn.LayerNorm(hidden_size, elementwise_affine=False, eps=1e-6)
        self.linear = nn.Linear(hidden_size, out_dim)
        self.adaLN_modulation = nn.Sequential(nn.SiLU(), nn.Linear(hidden_size, 2 * hidden_size))

    def forward(self, x: torch.Tensor, t: torch.Tensor):
        shift, scale = self.adaLN_modulation(t).chunk(2, dim=1)
        return self.linear(modulate(self.norm_final(x), shift, scale))


class DiT1d(BaseNNDiffusion):
    def __init__(self, in_dim: int, emb_dim: int, d_model: int = 384, n_heads: int = 6, depth: int = 12,
                 dropout: float = 0.0, timestep_emb_type: str = "positional",
                 timestep_emb_params: Optional[dict] = None):
        super().__init__(emb_dim, timestep_emb_type, timestep_emb_params)
        self.in_dim, self.emb_dim, self.d_model = in_dim, emb_dim, d_model
        self.x_proj = nn.Linear(in_dim, d_model)
        self.map_emb = nn.Sequential(nn.Linear(emb_dim, d_model), nn.Mish(), nn.Linear(d_model, d_model), nn.Mish())
        self.pos_emb = SinusoidalEmbedding(d_model)
        self.pos_emb_cache = None
        self.blocks = nn.ModuleList([DiTBlock(d_model, n_heads, dropout) for _ in range(depth)])
        self.final_layer = FinalLayer1d(d_model, in_dim)
        self.initialize_weights()

    def initialize_weights(self):
        """Xavier for Linears, N(0, 0.02) for the embedding MLP, zeros for every adaLN gate and the output head
        (so a fresh DiT1d outputs exactly 0 -- SURVEY Q14)."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
        for lin in (self.map_emb[0], self.map_emb[2]):
            nn.init.normal_(lin.weight, std=0.02)
        zeroed = [blk.adaLN_modulation[-1] for blk in self.blocks] + \
                 [self.final_layer.adaLN_modulation[-1], self.final_layer.linear]
        for lin in zeroed:
            nn.init.constant_(lin.weight, 0)
            nn.init.constant_(lin.bias, 0)

    def _tokens(self, x):
        if self.pos_emb_cache is None or self.pos_emb_cache.shape[0] != x.shape[1]:
            self.pos_emb_cache = self.pos_emb(torch.arange(x.shape[1], device=x.device))
        return self.x_proj(x) + self.pos_emb_cache[None, ]

    def _embed(self, noise, condition, zeros_if_none=True):
        emb = self.map_noise(noise)
        if condition is not None:
            emb = emb + condition
        elif zeros_if_none:
            emb = emb + torch.zeros_like(emb)
        return self.map_emb(emb)

    def forward(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor] = None):
        """x (b, horizon, in_dim), noise (b,), condition (b, emb_dim)|None -> (b, horizon, in_dim)."""
        from ..engine import train
        if train.supports_dit(self, x, condition):
            # autograd on, ROCm device: loss() / update() of DiT1d, and the classifier gradient through a HalfDiT1d trunk
            # (reference nn_classifier/half_dit.py:9, classifier/base.py:74-79) -- engine/train.py:dit_forward
            return train.dit_forward(self, x, noise, condition)
        if type(self) is DiT1d:
            from ..engine import dispatch
            y = dispatch.try_backbone_forward(self, x, noise, condition)     # GEMM/LN/attention launches on a ROCm device
            if y is not None:
                return y
        return self._forward_torch(x, noise, condition)

    def _forward_torch(self, x, noise, condition=None):
        h = self._tokens(x)
        emb = self._embed(noise, condition)
        for block in self.blocks:
            h = block(h, emb)
        return self.final_layer(h, emb)


class DiT1Ref(DiT1d):
    """DiT1d with a cross-attention to a reference trajectory carried in the first half of the feature axis."""

    def __init__(self, in_dim: int, emb_dim: int, d_model: int = 384, n_heads: int = 6, depth: int = 12,
                 dropout: float = 0.0, timestep_emb_type: str = "positional"):
        super().__init__(in_dim, emb_dim, d_model, n_heads, depth, dropout, timestep_emb_type)
        self.cross_attns = nn.ModuleList([nn.MultiheadAttention(d_model, n_heads, batch_first=True)
                                          for _ in range(depth)])

    def forward(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor] = None):
        """x (b, horizon, 2*in_dim) = [reference | noisy] -> (b, horizon, 2*in_dim) = [reference | prediction]."""
        if type(self) is DiT1Ref:
            from ..engine import dispatch
            y = dispatch.try_backbone_forward(self, x, noise, condition)     # cdx_dit1d_run with the cross-attention weights
            if y is not None:
                return y
        x_ref, x_cur = torch.chunk(x, 2, -1)
        keep = x_ref.clone()
        ref_tok = self._tokens(x_ref)
        h = self._tokens(x_cur)
        emb = self._embed(noise, condition, zeros_if_none=False)
        for cross, block in zip(self.cross_attns, self.blocks):
            h, _ = cross(h, ref_tok, ref_tok)
            h = block(h, emb)
        return torch.cat([keep, self.final_layer(h, emb)], -1)
