"""HalfJannerUNet1d -- the encoder half of the temporal U-Net with an MLP head, used as Diffuser's value/return
classifier (reference nn_classifier/half_jannerunet.py:11-125).  Same parameter names (``downs.{i}.{0,1,2}``,
``mid_block{1,2}.{0,1}``, ``final_block.{0,2}``) so reference classifier checkpoints load unchanged.

On a ROCm device with gradients off the forward (the ``log_p`` every ``sample()`` call ends with, reference
diffusionsde.py:597-601) is one launch of the same program kernel as the denoiser, per-step classifier *guidance* (d logp / dx) runs
inside the guided launch (engine/guided.py), and with autograd ON -- the classifier's own ``loss()`` / ``update()`` -- every
convolution, GroupNorm and Linear node is a library kernel forward and backward (engine/train.py:half_janner_forward).
"""
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from ..nn_diffusion.base_nn_diffusion import BaseNNDiffusion
from ..nn_diffusion.jannerunet import Downsample1d, ResidualBlock


class HalfJannerUNet1d(BaseNNDiffusion):
    def __init__(self, horizon: int, in_dim: int, out_dim: int = 1, kernel_size: int = 3, model_dim: int = 32,
                 emb_dim: int = 32, dim_mult: Tuple[int] = (1, 2, 2, 2), timestep_emb_type: str = "positional",
                 norm_type: str = "groupnorm"):
        super().__init__(emb_dim, timestep_emb_type)
        self.horizon, self.in_dim, self.out_dim = horizon, in_dim, out_dim
        self.kernel_size, self.model_dim, self.emb_dim, self.norm_type = kernel_size, model_dim, emb_dim, norm_type
        widths = [in_dim] + [int(model_dim * m) for m in np.cumprod(dim_mult)]
        stages = list(zip(widths[:-1], widths[1:]))

        self.map_emb = nn.Sequential(nn.Linear(emb_dim, model_dim * 4), nn.Mish(), nn.Linear(model_dim * 4, model_dim))
        self.downs = nn.ModuleList([])
        self.ups = nn.ModuleList([])              # kept (empty) for state_dict/attribute parity with the reference
        length = horizon
        for k, (ci, co) in enumerate(stages):
            last = k >= len(stages) - 1
            self.downs.append(nn.ModuleList([
                ResidualBlock(ci, co, model_dim, kernel_size, norm_type),
                ResidualBlock(co, co, model_dim, kernel_size, norm_type),
                nn.Identity() if last else Downsample1d(co)]))
            if not last:
                length //= 2
        top = widths[-1]
        self.mid_block1 = nn.ModuleList([ResidualBlock(top, top // 2, model_dim, kernel_size=5, norm_type=norm_type),
                                         Downsample1d(top // 2)])
        length //= 2
        self.mid_block2 = nn.ModuleList([ResidualBlock(top // 2, top // 4, model_dim, kernel_size=5, norm_type=norm_type),
                                         Downsample1d(top // 4)])
        length //= 2
        fc_dim = (top // 4) * max(length, 1)
        self.final_block = nn.Sequential(nn.Linear(fc_dim + model_dim, fc_dim // 2), nn.Mish(),
                                         nn.Linear(fc_dim // 2, out_dim))

    def _forward_torch(self, x, noise, condition):
        x = x.permute(0, 2, 1)
        emb = self.map_noise(noise)
        if condition is not None:
            emb = emb + condition
        emb = self.map_emb(emb)
        for res1, res2, down in self.downs:
            x = down(res2(res1(x, emb), emb))
        x = self.mid_block1[1](self.mid_block1[0](x, emb))
        x = self.mid_block2[1](self.mid_block2[0](x, emb))
        return self.final_block(torch.cat([x.flatten(1), emb], dim=-1))

    def forward(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor] = None):
        from ..engine import dispatch, train
        if train.supports_half_janner(self, x, condition):
            # autograd on, ROCm device (the classifier's loss() / update()): the same graph on the library's conv / GroupNorm / Linear nodes
            return train.half_janner_forward(self, x, noise, condition)
        y = dispatch.try_backbone_forward(self, x, noise, condition)
        return y if y is not None else self._forward_torch(x, noise, condition)
