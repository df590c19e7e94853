"""ChiUNet1d -- FiLM-conditioned temporal U-Net of Diffusion Policy (interface/checkpoint contract: reference
nn_diffusion/chiunet.py:13-192).  ``state_dict`` keys match (``downs.{i}.{0,1}.{conv1,conv2,cond_encoder.1,residual_conv}``,
``mids.{0,1}``, ``ups``, ``global_cond_encoder`` / ``local_cond_encoder``, ``final_conv``).

Execution: this nn.Module is the parameter container and the PyTorch (CPU / autograd / local-conditioning) path.  With
``obs_as_global_cond=True`` on a ROCm device the forward -- and, through ``DDPM.sample`` / ``DiscreteDiffusionSDE.sample``,
the whole denoising loop -- runs natively: nets below 10 M parameters at small batch in the fused program kernel
(engine/program2.py:compile_chiunet2: FiLM is an epilogue of the first conv of each block, the blocks' FiLM vectors are rows of a
per-(step, trajectory) table), everything else -- BASELINE config 3 (68.9 M parameters) at every batch -- on the implicit-GEMM
executor (engine/bigbatch.py, ``cdx_chiunet_run``).
"""
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn

from ..utils import GroupNorm1d
from .base_nn_diffusion import BaseNNDiffusion
from .jannerunet import Downsample1d, Upsample1d


def _cna(c_in, c_out, k):
    return nn.Sequential(nn.Conv1d(c_in, c_out, k, padding=k // 2), GroupNorm1d(c_out, 8, 4), nn.Mish())


class ChiResidualBlock(nn.Module):
    """CNA -> FiLM(emb) -> CNA, plus a 1x1 / identity skip.  FiLM = per-channel (scale, bias) when
    ``cond_predict_scale`` else an additive bias."""

    def __init__(self, in_dim: int, out_dim: int, emb_dim: int, kernel_size: int = 3, cond_predict_scale: bool = False):
        super().__init__()
        self.conv1 = _cna(in_dim, out_dim, kernel_size)
        self.conv2 = _cna(out_dim, out_dim, kernel_size)
        self.cond_predict_scale = cond_predict_scale
        self.out_dim = out_dim
        self.cond_encoder = nn.Sequential(nn.Mish(), nn.Linear(emb_dim, 2 * out_dim if cond_predict_scale else out_dim))
        self.residual_conv = nn.Conv1d(in_dim, out_dim, 1) if in_dim != out_dim else nn.Identity()

    def forward(self, x, emb):
        h = self.conv1(x)
        film = self.cond_encoder(emb)
        if self.cond_predict_scale:
            film = film.reshape(film.shape[0], 2, self.out_dim, 1)
            h = film[:, 0, ...] * h + film[:, 1, ...]
        else:
            h = h + film.unsqueeze(-1)
        return self.conv2(h) + self.residual_conv(x)


class ChiUNet1d(BaseNNDiffusion):
    def __init__(self, act_dim: int, obs_dim: int, To: int, model_dim: int = 256, emb_dim: int = 256,
                 kernel_size: int = 5, cond_predict_scale: bool = True, obs_as_global_cond: bool = True,
                 dim_mult: List[int] = [1, 2, 2], timestep_emb_type: str = "positional",
                 timestep_emb_params: Optional[dict] = None):
        super().__init__(emb_dim, timestep_emb_type, timestep_emb_params)
        self.obs_as_global_cond, self.model_dim, self.emb_dim = obs_as_global_cond, model_dim, emb_dim
        widths = [act_dim] + [int(model_dim * m) for m in np.cumprod(dim_mult)]
        stages = list(zip(widths[:-1], widths[1:]))

        self.map_emb = nn.Sequential(nn.Linear(emb_dim, emb_dim * 4), nn.Mish(), nn.Linear(emb_dim * 4, emb_dim))
        film_dim = emb_dim

        def block(ci, co):
            return ChiResidualBlock(ci, co, film_dim, kernel_size, cond_predict_scale)

        if obs_as_global_cond:
            self.global_cond_encoder = nn.Linear(To * obs_dim, emb_dim)
            film_dim = emb_dim * 2                      # FiLM input = [time emb | encoded obs]
            self.local_cond_encoder = None
        else:
            self.global_cond_encoder = None
            self.local_cond_encoder = nn.ModuleList([block(obs_dim, model_dim), block(obs_dim, model_dim),
                                                     Downsample1d(model_dim)])
        n_res = len(stages)
        self.downs = nn.ModuleList([
            nn.ModuleList([block(ci, co), block(co, co), Downsample1d(co) if k < n_res - 1 else nn.Identity()])
            for k, (ci, co) in enumerate(stages)])
        self.ups = nn.ModuleList([])       # registered before the mid blocks, like the reference (parameter order)
        top = widths[-1]
        self.mids = nn.ModuleList([block(top, top), block(top, top)])
        self.ups.extend([
            nn.ModuleList([block(co * 2, ci), block(ci, ci), Upsample1d(ci) if k < n_res - 1 else nn.Identity()])
            for k, (ci, co) in enumerate(reversed(stages[1:]))])
        self.final_conv = nn.Sequential(_cna(model_dim, model_dim, kernel_size)[0], GroupNorm1d(model_dim, 8, 4),
                                        nn.Mish(), nn.Conv1d(model_dim, act_dim, 1))

    def forward(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor] = None):
        """x (b, Ta, act_dim), noise (b,), condition (b, To, obs_dim) [required, SURVEY Q12] -> (b, Ta, act_dim)."""
        assert x.shape[1] & (x.shape[1] - 1) == 0, "Ta dimension must be 2^n"
        from ..engine import dispatch, train
        if train.supports_chi(self, x, condition):
            # autograd on, ROCm device (loss() / update()): the same graph, every convolution / GroupNorm / FiLM Linear node on the
            # library's kernels forward and backward (engine/train.py)
            return train.chi_forward(self, x, noise, condition)
        y = dispatch.try_backbone_forward(self, x, noise, condition)      # one fused launch on a ROCm device
        if y is not None:
            return y
        return self._forward_torch(x, noise, condition)

    def _forward_torch(self, x: torch.Tensor, noise: torch.Tensor, condition: Optional[torch.Tensor] = None):
        x = x.permute(0, 2, 1)
        emb = self.map_emb(self.map_noise(noise))
        local = None
        if self.obs_as_global_cond:
            emb = torch.cat([emb, self.global_cond_encoder(torch.flatten(condition, 1))], dim=-1)
        else:
            c = condition.permute(0, 2, 1)
            assert x.shape[-1] == c.shape[-1]
            enc1, enc2, enc_down = self.local_cond_encoder
            local = [enc1(c, emb), enc_down(enc2(c, emb))]
        skips = []
        for idx, (res1, res2, down) in enumerate(self.downs):
            x = res1(x, emb)
            if idx == 0 and local is not None:
                x = x + local[0]
            x = res2(x, emb)
            skips.append(x)
            x = down(x)
        for mid in self.mids:
            x = mid(x, emb)
        for idx, (res1, res2, up) in enumerate(self.ups):
            x = res1(torch.cat((x, skips.pop()), dim=1), emb)
            if idx == len(self.ups) - 1 and local is not None:
                x = x + local[1]
            x = up(res2(x, emb))
        return self.final_conv(x).permute(0, 2, 1)
