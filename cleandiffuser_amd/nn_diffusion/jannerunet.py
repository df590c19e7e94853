"""JannerUNet1d -- temporal 1-D U-Net denoiser (Diffuser).

Mirror of reference nn_diffusion/jannerunet.py:12-201 at the *interface* level: same constructor,
same ``state_dict`` keys (``downs.{i}.{0,1}.conv{1,2}.{0,1}``, ``emb_mlp.1``, ``residual_conv``,
``downs.{i}.3.conv``, ``ups.{i}.3.conv``, ``mid_block{1,2}``, ``final_conv.{0,1,3}``, ``map_emb.{0,2}``)
so reference checkpoints load unchanged.

Execution: this nn.Module is the *parameter container + CPU path*.  On a ROCm device with autograd ON (``loss()`` /
``update()``) the forward and backward of every convolution and GroupNorm are library kernels behind
``torch.autograd.Function`` nodes (`cleandiffuser_amd.engine.train`); with
``requires_grad`` off, ``forward`` is served by the fused gfx950 program kernel
(`cleandiffuser_amd.engine`, C-ABI ``cdx_unet2_run``) -- the whole U-Net forward in ONE launch with
activations resident in LDS -- and ``DiscreteDiffusionSDE.sample`` goes one step further and runs
the entire denoising loop inside that launch.
"""
from typing import List, Optional

import einops
import numpy as np
import torch
import torch.nn as nn

from ..utils import GroupNorm1d
from .base_nn_diffusion import BaseNNDiffusion


class LayerNorm(nn.Module):
    """Channel LayerNorm over dim=1 of (b, C, L) (reference jannerunet.py:37-48)."""

    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.eps = eps
        self.g = nn.Parameter(torch.ones(1, dim, 1))
        self.b = nn.Parameter(torch.zeros(1, dim, 1))

    def forward(self, x):
        mean = torch.mean(x, dim=1, keepdim=True)
        var = torch.var(x, dim=1, unbiased=False, keepdim=True)
        return (x - mean) / (var + self.eps).sqrt() * self.g + self.b


def get_norm(dim: int, norm_type: str = "groupnorm"):
    if norm_type == "groupnorm":
        return GroupNorm1d(dim, 8, 4)
    if norm_type == "layernorm":
        return LayerNorm(dim)
    return nn.Identity()


class Downsample1d(nn.Module):
    """Conv1d(C, C, k=3, stride=2, pad=1): L -> L/2."""

    def __init__(self, dim):
        super().__init__()
        self.conv = nn.Conv1d(dim, dim, 3, 2, 1)

    def forward(self, x):
        return self.conv(x)


class Upsample1d(nn.Module):
    """ConvTranspose1d(C, C, k=4, stride=2, pad=1): L -> 2L."""

    def __init__(self, dim):
        super().__init__()
        self.conv = nn.ConvTranspose1d(dim, dim, 4, 2, 1)

    def forward(self, x):
        return self.conv(x)


def _conv_norm_act(c_in, c_out, k, norm_type):
    return nn.Sequential(nn.Conv1d(c_in, c_out, k, padding=k // 2), get_norm(c_out, norm_type), nn.Mish())


class ResidualBlock(nn.Module):
    """y = CNA2(CNA1(x) + Linear(Mish(emb))[:, :, None]) + skip(x),  CNA = Conv1d -> Norm -> Mish."""

    def __init__(self, in_dim: int, out_dim: int, emb_dim: int, kernel_size: int = 3, norm_type: str = "groupnorm"):
        super().__init__()
        self.conv1 = _conv_norm_act(in_dim, out_dim, kernel_size, norm_type)
        self.conv2 = _conv_norm_act(out_dim, out_dim, kernel_size, norm_type)
        self.emb_mlp = nn.Sequential(nn.Mish(), nn.Linear(emb_dim, out_dim))
        self.residual_conv = nn.Conv1d(in_dim, out_dim, 1) if in_dim != out_dim else nn.Identity()

    def forward(self, x, emb):
        h = self.conv1(x) + self.emb_mlp(emb).unsqueeze(-1)
        return self.conv2(h) + self.residual_conv(x)


class LinearAttention(nn.Module):
    """O(L) attention (reference jannerunet.py:72-95); off in every shipped pipeline, PyTorch path only."""

    def __init__(self, dim, heads=4, dim_head=32):
        super().__init__()
        self.norm = LayerNorm(dim)
        self.scale = dim_head ** -0.5
        self.heads = heads
        inner = dim_head * heads
        self.to_qkv = nn.Conv1d(dim, inner * 3, 1, bias=False)
        self.to_out = nn.Conv1d(inner, dim, 1)

    def forward(self, x):
        x = self.norm(x)
        q, k, v = (einops.rearrange(t, "b (h c) d -> b h c d", h=self.heads) for t in self.to_qkv(x).chunk(3, dim=1))
        q = q * self.scale
        k = k.softmax(dim=-1)
        ctx = torch.einsum("b h d n, b h e n -> b h d e", k, v)
        out = torch.einsum("b h d e, b h d n -> b h e n", ctx, q)
        out = einops.rearrange(out, "b h c d -> b (h c) d")
        return self.to_out(out) + x


class JannerUNet1d(BaseNNDiffusion):
    def __init__(self, in_dim: int, model_dim: int = 32, emb_dim: int = 32, kernel_size: int = 3,
                 dim_mult: List[int] = [1, 2, 2, 2], norm_type: str = "groupnorm", attention: bool = False,
                 timestep_emb_type: str = "positional", timestep_emb_params: Optional[dict] = None):
        super().__init__(emb_dim, timestep_emb_type, timestep_emb_params)
        self.in_dim, self.model_dim, self.emb_dim = in_dim, model_dim, emb_dim
        self.kernel_size, self.norm_type, self.attention = kernel_size, norm_type, attention

        widths = [in_dim] + [int(model_dim * m) for m in np.cumprod(dim_mult)]
        stages = list(zip(widths[:-1], widths[1:]))
        n_res = len(stages)

        def block(ci, co):
            return ResidualBlock(ci, co, model_dim, kernel_size, norm_type)

        def attn(c):
            return LinearAttention(c) if attention else nn.Identity()

        self.map_emb = nn.Sequential(nn.Linear(emb_dim, model_dim * 4), nn.Mish(), nn.Linear(model_dim * 4, model_dim))

        self.downs = nn.ModuleList([
            nn.ModuleList([block(ci, co), block(co, co), attn(co),
                           Downsample1d(co) if k < n_res - 1 else nn.Identity()])
            for k, (ci, co) in enumerate(stages)])

        self.ups = nn.ModuleList([])       # registered before the mid blocks, like the reference (parameter order)
        top = widths[-1]
        self.mid_block1 = block(top, top)
        self.mid_attn = attn(top)
        self.mid_block2 = block(top, top)

        # NB: the loop index never reaches n_res-1, so every up stage upsamples and the
        # first skip (downs[0] output) is never consumed.
        self.ups.extend([
            nn.ModuleList([block(co * 2, ci), block(ci, ci), attn(ci),
                           Upsample1d(ci) if k < n_res - 1 else nn.Identity()])
            for k, (ci, co) in enumerate(reversed(stages[1:]))])

        self.final_conv = nn.Sequential(
            nn.Conv1d(model_dim, model_dim, 5, padding=2), get_norm(model_dim, norm_type), nn.Mish(),
            nn.Conv1d(model_dim, in_dim, 1))

    # ------------------------------------------------------------------ #
    def _forward_torch(self, x, noise, condition):
        """Plain PyTorch path (CPU, autograd, unsupported variants)."""
        x = x.permute(0, 2, 1)
        emb = self.map_noise(noise)
        emb = emb + (condition if condition is not None else torch.zeros_like(emb))
        emb = self.map_emb(emb)

        skips = []
        for res1, res2, att, down in self.downs:
            x = att(res2(res1(x, emb), emb))
            skips.append(x)
            x = down(x)
        x = self.mid_block2(self.mid_attn(self.mid_block1(x, emb)), emb)
        for res1, res2, att, up in self.ups:
            x = torch.cat([x, skips.pop()], dim=1)
            x = up(att(res2(res1(x, emb), emb)))
        return self.final_conv(x).permute(0, 2, 1)

    def forward(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor] = None):
        """x (b, horizon, in_dim), noise (b,), condition (b, emb_dim)|None -> (b, horizon, in_dim)."""
        assert x.shape[1] & (x.shape[1] - 1) == 0, "Ta dimension must be 2^n"
        from ..engine import dispatch, train
        if train.supports(self, x, condition):
            # autograd on, ROCm device (loss() / update(), sampling with requires_grad=True): the same graph, every convolution and
            # GroupNorm node on the library's kernels forward and backward (engine/train.py)
            return train.janner_forward(self, x, noise, condition)
        y = dispatch.try_backbone_forward(self, x, noise, condition)
        if y is not None:
            return y
        return self._forward_torch(x, noise, condition)
